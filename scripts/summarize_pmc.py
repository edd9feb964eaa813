"""Summarise rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in separate runs, as MI355X_MICROARCH.md's HBM
section prescribes) into HBM bytes per launch for the GEMM kernels.  gfx950 correction: FETCH_SIZE counts 128-B
requests as 64 B for wide coalesced streams -> doubled; units are KiB."""
import collections, csv, json, re, sys

def agg(path, name):
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != name:
            continue
        kn = r['Kernel_Name']
        # every NT launch (single problem, batched, bf16 storage) / every weight-gradient launch (single, batched, both tile sizes)
        k = ('gemm_nt_kernel' if re.search(r'gemm_nt(2b?|16b?|6)?_kernel', kn) else
             'gemm_tn_kernel' if re.search(r'gemm_tn(2b?|b|16|16x256|256)?_kernel', kn) else
             'calib_read4' if 'calib_read4' in kn else 'calib_read16' if 'calib_read16' in kn else None)
        if k:
            out[k][0] += 1
            out[k][1] += float(r['Counter_Value'])
    return out

fetch_csv, write_csv, out_json = sys.argv[1:4]
bench_json = sys.argv[4] if len(sys.argv) > 4 else None      # the bench line printed by the profiled run: records the workload
f, w = agg(fetch_csv, 'FETCH_SIZE'), agg(write_csv, 'WRITE_SIZE')
res = {}
if bench_json:
    b = json.loads(open(bench_json).read().strip().splitlines()[-1])
    c = b['config']
    # what bench.py matches a later run against before it quotes these numbers as that run's `roofline.traffic`
    res['workload_record'] = {"rays": c['rays_per_gpu'], "real_capture": 'real-capture (is_nerf' in c['workload'], "mlp_dtype": 'bf16x6' if b['dtype'].startswith('bf16x6') else {'f32': 'fp32', 'bf16': 'bf16'}.get(b['dtype'], b['dtype']),
                              "bf16_storage": b['dtype'] == 'bf16', "mean_inner_points": c['mean_inner_points'],
                              "mean_outer_points": c['mean_outer_points'], "steps": b['steps'], "warmup": b['warmup']}
for k in ('gemm_nt_kernel', 'gemm_tn_kernel'):          # launches per step: a later run must have the same launch structure to quote these per-launch bytes
    if bench_json and k in f:
        res['workload_record'][k.replace('gemm_', '').replace('_kernel', '') + '_launches_per_step'] = f[k][0] / float(b['steps'] + b['warmup'])
# bench.py times a batch of problems launched back to back (nt_batch outside the exact-fp32 mode) as ONE launch: the per-launch
# bytes above are per KERNEL; the event-launch counts of the stats pass let bench.py rescale them to its own launch unit
if bench_json and len(sys.argv) > 5:
    bs = json.loads(open(sys.argv[5]).read().strip().splitlines()[-1]).get('roofline', {})
    if bs.get('launches'):
        res['workload_record']['nt_event_launches_per_step'] = float(bs['launches'])
    if bs.get('wgrad', {}).get('launches'):
        res['workload_record']['tn_event_launches_per_step'] = float(bs['wgrad']['launches'])
for k in f:
    n = f[k][0]
    res[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * 1024 * f[k][1] / n, "write_bytes_per_launch": 1024 * w[k][1] / max(w[k][0], 1),
              "note": "FETCH_SIZE x2 (gfx950 128-B request correction), KiB units; separate --pmc passes"}
    res[k]["hbm_bytes_per_launch"] = res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
    if k == 'gemm_tn_kernel':
        res[k]["note"] += ("; all weight-gradient launches (gemm_tn2_kernel / gemm_tn256_kernel + gemm_tn_kernel); the x2 factor was calibrated for 4-byte-"
                           "per-lane reads too (scripts/gemm_lab calib)")
json.dump(res, open(out_json, 'w'), indent=1)
print(json.dumps(res, indent=1))
