import os, sys, ctypes as C, tempfile, numpy as np
sys.path.insert(0, os.getcwd())
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes
w,h=1920,1080
path=scenes.write_scene(scenes.stress_scene_text((22,22,21),res=(w,h),depth=8), os.path.join(tempfile.mkdtemp(),'s.txt'))
sc=capi.Scene(path,res=(w,h))
lib=C.CDLL(os.environ['PT_AMD_LIB'])
out=(C.c_ulonglong*8)()
r=capi.Renderer(sc, arith='fast')
lib.pt_grid_stats_fast(out,1)
r.render(1,20); r.sync()
lib.pt_grid_stats_fast(out,0)
v=np.array(list(out),float); g=v[7]
print('groups',g,'steps/group %.1f on-lanes/step %.1f rounds/group %.1f filter chunks/group %.2f items/group %.0f prim chunks/group %.2f cands/group %.0f'%(v[0]/g, v[6]/v[0], v[1]/g, v[2]/g, v[3]/g, v[4]/g, v[5]/g))
r.free()
