set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=gpurun_out/s3_g; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 500 python3 tools/scene_ladder.py --sizes 500,1000,c5 --arith fast --spp 100 --flags 0,1536,4608 > $O/ladder.log 2>&1; cat $O/ladder.log
