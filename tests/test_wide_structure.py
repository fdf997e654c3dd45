"""Host-side checks of the 4-wide surface-area tree (pt_build_wide, csrc/pt_api.cpp build_wide): the properties the
"same image with any structure" argument of csrc/pt_wide.inc rests on — every leaf of the reference's tree appears exactly
once with ITS box bit for bit and its visiting-order index, every inner box is the exact union of its node's children, the
links form a tree rooted at node 0 — plus the stack bound the kernel's LDS is sized with.  No GPU needed."""
import numpy as np
import pytest

from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes


def thread(bvh):
    """Visiting order of the reference's stack walk (node, right subtree, left subtree): reference index -> threaded index."""
    order, stack = [], [0]
    while stack:
        i = stack.pop()
        order.append(i)
        if bvh[i].left >= 0:
            stack.append(bvh[i].left)
            stack.append(bvh[i].right)
    return {ref: t for t, ref in enumerate(order)}


def walk_checks(info, boxes, links, leaf_boxes):
    n = len(links)
    seen_leaf, parent_box = {}, {0: None}
    visited = set()

    def depth_need(w):  # stack entries a walk can hold below node w
        assert w not in visited, "a node has two parents"
        visited.add(w)
        inner, lo, hi = [], np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)
        for j in range(4):
            l = int(links[w, j])
            if l == 0:
                continue
            b = boxes[w, :, j]
            lo, hi = np.minimum(lo, b[:3]), np.maximum(hi, b[3:])
            if l < 0:
                u = l & 0xFFFFFFFF
                leaf, geom, typ = u & 0x7FFF, (u >> 15) & 0x3FFF, (u >> 29) & 3
                assert leaf not in seen_leaf
                seen_leaf[leaf] = (geom, typ)
                assert np.array_equal(b.view(np.uint32), leaf_boxes[leaf].view(np.uint32)), "a leaf's box is the reference's, bit for bit"
            else:
                assert 0 < l < n
                parent_box[l] = b.copy()
                inner.append(l)
        if parent_box.get(w) is not None:
            assert np.array_equal(np.concatenate([lo, hi]).view(np.uint32), parent_box[w].view(np.uint32)), "an inner box is the exact union of its children"
        need = [depth_need(c) for c in inner]
        return 0 if not inner else len(inner) - 1 + max(need)

    need = depth_need(0)
    assert visited == set(range(n))
    assert max(1, need) == info.max_stack
    return seen_leaf


@pytest.mark.parametrize("prims,clustered", [(1, False), (2, False), (7, False), (33, False), (156, False), (156, True), (1000, False), (3000, True)])
def test_wide_tree_structure(tmp_path, prims, clustered):
    res = (64, 48)
    if prims == 7:
        text = scenes.cornell_scene_text(res=res)
    elif prims <= 2:
        text = scenes.random_scene_text(5, 0, res=res)  # the six walls ...
        text = "".join(text.split("OBJECT 1")[:1]) if prims == 1 else "".join(text.split("OBJECT 2")[:1])  # ... cut down to one / two objects
    else:
        text = scenes.random_scene_text(40 + prims, prims - 6, res=res, clustered=clustered)
    sc = capi.Scene(scenes.write_scene(text, str(tmp_path / "s.txt")), res=res)
    assert sc.desc.num_geoms == prims
    bvh = sc.bvh()
    tmap = thread(bvh)
    leaf_boxes = {tmap[i]: np.array(list(nd.bmin) + list(nd.bmax), np.float32) for i, nd in enumerate(bvh) if nd.left < 0}
    leaf_geom = {tmap[i]: nd.geomIndex for i, nd in enumerate(bvh) if nd.left < 0}
    info, boxes, links = sc.wide_tree()
    assert info.num_nodes == len(links) >= 1 and info.num_leaves == prims
    seen = walk_checks(info, boxes, links, leaf_boxes)
    assert set(seen) == set(leaf_boxes), "every leaf of the reference's tree exactly once"
    for leaf, (geom, typ) in seen.items():
        assert geom == leaf_geom[leaf] and typ == sc.desc.geoms[geom].type
    # four children per node pay: far fewer nodes than the binary tree has inner nodes
    if prims >= 33:
        assert info.num_nodes <= (prims - 1) * 0.6
        assert info.max_stack <= 48


def test_wide_tree_with_tightened_boxes(tmp_path):
    """The boxes pt_init uses from 64 BVH nodes on (sphere leaves tightened, pt_traversal_boxes) go into the leaf records unchanged."""
    res = (64, 48)
    sc = capi.Scene(scenes.write_scene(scenes.random_scene_text(11, 150, res=res), str(tmp_path / "s.txt")), res=res)
    tb, tight = sc.traversal_boxes()
    assert tight > 0
    bvh = sc.bvh()
    tmap = thread(bvh)
    leaf_boxes = {tmap[i]: tb[nd.geomIndex] for i, nd in enumerate(bvh) if nd.left < 0}
    info, boxes, links = sc.wide_tree(tighten=True)
    walk_checks(info, boxes, links, leaf_boxes)
