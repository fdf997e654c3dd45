mkdir -p gpurun_out/r04bi
L=gpurun_out/r04bi/small_scan.log
echo "== default (LDS tables, all leaves in the top list)" > $L; python3 tools/scene_ladder.py --spp 50 --arith fast,exact --sizes 16,24,32 >> $L 2>&1
for top in 32 16 8 4; do echo "== scan mode, top $top" >> $L; PT_LDS_TABLE_KB=0 PT_TOP_ENTRIES=$top python3 tools/scene_ladder.py --spp 50 --arith fast,exact --sizes 16,24,32 >> $L 2>&1; done
cat $L
