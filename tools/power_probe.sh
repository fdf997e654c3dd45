#!/bin/bash
# usage: pw.sh LIB  -> runs bench 3000 steps with that lib while sampling rocm-smi power/clock
LIB=$1
if [ "$LIB" != "-" ]; then export PT_AMD_LIB=$(readlink -f $LIB); fi
python3 bench.py --no-extras --steps 4000 --warmup 100 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Graphics Package Power|Socket Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.3; done
wait $BP
python3 -c "import json; d=json.loads(open('/tmp/b.json').read().strip().split(chr(10))[-1]); print('$LIB', d['value'], d['roofline']['avg_launch_us'])"
