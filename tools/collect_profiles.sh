#!/bin/bash
# After `gpurun -- 'bash tools/profile_cfg4.sh r05_cfg4; bash tools/profile_bench.sh r05_bench; bash tools/prof_mixed_all.sh r05 pmc'`:
# copies the sha-locked records and summaries from gpurun_out/ (scratch) into profiles/ (tracked) and rebuilds profiles/<tag>_mixed_summary.txt.
# bench.py only attaches records whose source hash matches the tree, so this must run BEFORE the bench line that should carry them.
TAG=${1:-r06}
cd "$(dirname "$0")/.."
for f in ${TAG}_cfg4_valu.json ${TAG}_cfg4_summary.txt ${TAG}_cfg4_ksmac_counters.json ${TAG}_cfg4_tensor_bsk_counters.json ${TAG}_cfg4_tensor_q_counters.json ${TAG}_bench_ksmac_counters.json ${TAG}_bench_chain_valu.json ${TAG}_bench_summary.txt ${TAG}_bench_two_streams_summary.txt; do
  cp gpurun_out/$f profiles/$f
done
python3 - "$TAG" <<'PY'
import sys
tag = sys.argv[1]
out = open('profiles/%s_mixed_summary.txt' % tag).read().split('\n')[:5] + ['']
for n in ('8192', '16384'):
    out += ['# N=%s  %s' % (n, l.strip()) for l in open('gpurun_out/%s_mixed_%s.log' % (tag, n)) if 'ops/s' in l]
for n in ('8192', '16384'):
    out += ['', '################ N = %s: kernel trace' % n] + open('gpurun_out/%s_mixed_trace_%s.txt' % (tag, n)).read().rstrip().split('\n')
    out += ['', '################ N = %s: SQ counters' % n] + open('gpurun_out/%s_mixed_pmc_%s.txt' % (tag, n)).read().rstrip().split('\n')
open('profiles/%s_mixed_summary.txt' % tag, 'w').write('\n'.join(out) + '\n')
PY
python3 -c "
import json, bench
sha = bench.all_sources_sha()
for f in ('profiles/${TAG}_cfg4_valu.json', 'profiles/${TAG}_bench_chain_valu.json'):
    print(f, 'matches the tree' if json.load(open(f)).get('sources_sha') == sha else 'STALE')
"
