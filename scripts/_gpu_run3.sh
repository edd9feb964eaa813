cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_kernarg_ab.txt
: > $O
run() { # tag, env, flags
  local tag=$1; local envs=$2; shift 2
  local line=$(env $envs python bench.py --no-cpu-baseline --no-extra --no-kernel-timing "$@" 2>/dev/null | tail -1)
  python - "$tag" "$envs" "$line" >> $O <<'PY'
import sys, json
tag, envs, line = sys.argv[1:4]
d = json.loads(line)
print(f"{tag:14s} {envs:28s} {d['ms_per_step']:.3f} ms/step")
PY
}
for rep in 1 2; do
  run thick1024 "NU_AB=0" --workload stage2 --thick --rays 1024 --steps 30 --warmup 8
  run thick1024 "HIP_FORCE_DEV_KERNARG=1" --workload stage2 --thick --rays 1024 --steps 30 --warmup 8
  run thick1024 "HIP_FORCE_DEV_KERNARG=0" --workload stage2 --thick --rays 1024 --steps 30 --warmup 8
  run rays512 "NU_AB=0" --rays 512 --steps 40 --warmup 10
  run rays512 "HIP_FORCE_DEV_KERNARG=1" --rays 512 --steps 40 --warmup 10
  run rays512 "HIP_FORCE_DEV_KERNARG=0" --rays 512 --steps 40 --warmup 10
  run headline "NU_AB=0" --steps 20 --warmup 5
  run headline "HIP_FORCE_DEV_KERNARG=1" --steps 20 --warmup 5
  run headline "HIP_FORCE_DEV_KERNARG=0" --steps 20 --warmup 5
  cat $O | tail -9
done
