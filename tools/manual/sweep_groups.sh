# default bench shape (configs[3], 256 filters) against the number of stream groups of the large-state chain
for rep in 1 2 3; do
for g in ${GROUPS_LIST:-2 3 4}; do
  ASLAM_LARGE_GROUPS=$g timeout -k 10 150 python bench.py --no-sub --no-legs --cpu-sample 0 > gpurun_out/groups_$g.json 2>/dev/null
  python - $g $rep <<'PY'
import json,sys
g=sys.argv[1]
r=json.loads(open('gpurun_out/groups_%s.json'%g).read().strip().splitlines()[-1])
print('rep',sys.argv[2],'groups',g,'value %.0f'%r['value'],'ms_per_step %.2f'%r['ms_per_step'])
PY
done
done
