set -e
cd $GRAFT_REPO_ROOT
for i in 1 2; do
NU_TN_BATCH=0 NU_NT_BATCH=0 python bench.py --steps 30 --warmup 8 --no-extra --no-cpu-baseline > gpurun_out/r4_ab_nobatch_$i.json 2>gpurun_out/r4_ab_nobatch_$i.err
python bench.py --steps 30 --warmup 8 --no-extra --no-cpu-baseline > gpurun_out/r4_ab_new_$i.json 2>gpurun_out/r4_ab_new_$i.err
NU_TN_BATCH=0 python bench.py --steps 30 --warmup 8 --no-extra --no-cpu-baseline > gpurun_out/r4_ab_notnbatch_$i.json 2>gpurun_out/r4_ab_notnbatch_$i.err
done
for r in 512; do
NU_TN_BATCH=0 NU_NT_BATCH=0 python bench.py --rays $r --steps 40 --warmup 8 --no-extra --no-cpu-baseline > gpurun_out/r4_ab_nobatch_rays$r.json 2>/dev/null
python bench.py --rays $r --steps 40 --warmup 8 --no-extra --no-cpu-baseline > gpurun_out/r4_ab_new_rays$r.json 2>/dev/null
NU_TN_BATCH=0 python bench.py --rays $r --steps 40 --warmup 8 --no-extra --no-cpu-baseline > gpurun_out/r4_ab_notnbatch_rays$r.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        r=d.get('roofline',{})
        print(f, round(d['ms_per_step'],3), round(d['value']), r.get('frac'), r.get('launches'), d.get('wgrad',{}).get('achieved'), d.get('wgrad',{}).get('launches'))
    except Exception as e: print(f, 'ERR', e)
PY
