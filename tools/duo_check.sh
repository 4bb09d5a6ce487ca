#!/bin/bash
# GPU box: the two-waves-per-env one-step kernel (CZ_DUO=1, cz_duo.h) against the ordinary one, same library, same box:
# parity suites under CZ_DUO=1, launch time under three action streams, the bench's event-timed kernel, device-clock timelines.
# usage: bash tools/duo_check.sh TAG [skip-tests]
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
if [ -z "$2" ]; then
  CZ_DUO=1 timeout 1200 python -m pytest tests -m gpu -x -q -k "parity or rollout or api or fuzz or spawn or rotation or symbolic" 2>&1 | tail -8 > $O/duo_tests_$TAG.log
fi
for rep in 1 2; do
  for duo in 0 1; do
    echo "== CZ_DUO=$duo rep $rep"
    CZ_DUO=$duo timeout 200 python3 tools/mode_timing.py
  done
done > $O/duo_modes_$TAG.txt 2>&1
for duo in 0 1; do
  CZ_DUO=$duo CZ_CHAIN=0 timeout 600 python3 bench.py --steps 2000 --warmup 200 --repeats 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('CZ_DUO=$duo ordered kernel_us', d['roofline']['kernel_us'], 'value', d['value']/1e6, 'cook', d.get('cooking_policy',{}).get('us_per_launch'), 'closed', d.get('closed_loop',{}).get('us_per_step'))
"
done > $O/duo_bench_$TAG.txt 2>&1
for duo in 0 1; do
  CZ_DUO=$duo timeout 300 python3 tools/timeline.py 4096 2000 0 > $O/duo_timeline_${duo}_$TAG.json 2>$O/duo_timeline_${duo}_$TAG.err
  CZ_DUO=$duo timeout 300 python3 tools/timeline.py 4096 2000 0 stay > $O/duo_timeline_stay_${duo}_$TAG.json 2>>$O/duo_timeline_${duo}_$TAG.err
done
tail -n 30 $O/duo_tests_$TAG.log $O/duo_modes_$TAG.txt $O/duo_bench_$TAG.txt
for duo in 0 1; do python3 -c "
import json
for f in ('$O/duo_timeline_${duo}_$TAG.json','$O/duo_timeline_stay_${duo}_$TAG.json'):
    try:
        d=json.load(open(f)); print(f, 'events', round(d['hip_events_us_per_launch'],3), 's2s', d['start_to_start_us']['median'], 'gap', d['gap_last_out_to_next_first_in_us']['median'], 'life', d['wave_lifetime_us'], 'starts', d['wave_start_offset_us_percentiles']['99'])
    except Exception as e: print(f, 'ERR', e)
"; done
