#!/bin/bash
# GPU box: device-clock timelines (tools/timeline.py, timeline library) of the ordinary one-step kernel and of the
# two-waves-per-env experiment's variants (CZ_DUO=1 CZ_STOP=v)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for cfg in "0 -1" "1 0" "1 3" "1 4"; do
  set -- $cfg
  for mode in random stay; do
    CZ_DUO=$1 CZ_STOP=$2 timeout 300 python3 tools/timeline.py 4096 2000 0 $mode > $O/duo_tl_duo$1_v$2_$mode.json 2> $O/duo_tl.err
    python3 -c "
import json
d=json.load(open('$O/duo_tl_duo$1_v$2_$mode.json'))
print('CZ_DUO=$1 variant $2 $mode: events', round(d['hip_events_us_per_launch'],3), 'start-to-start', d['start_to_start_us']['median'], 'boundary gap', d['gap_last_out_to_next_first_in_us']['median'], 'all waves started', d['wave_start_offset_us_percentiles']['99'], 'wave lifetime', d['wave_lifetime_us'])
"
  done
done | tee $O/duo_timelines.txt
