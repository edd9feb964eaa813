"""Summarise a rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY,
SQ_ACTIVE_INST_ANY; collected with --kernel-trace) for the two GEMM kernels: effective shader clock under the kernel
(GRBM_GUI_ACTIVE / 8 XCDs / duration, MI355X_MICROARCH.md 'DVFS give-back'), matrix-pipe busy fraction at that clock
(MFMA busy cycles / (1024 SIMDs x cycles)), and where the waves' cycles go."""
import re, collections, csv, json, sys

cc_csv, kt_csv, out_json = sys.argv[1:4]
dur = {}
for r in csv.DictReader(open(kt_csv)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(cc_csv)):
    name = r['Kernel_Name']
    k = 'gemm_nt_kernel' if re.search(r'gemm_nt(2b?|6|16b?)?_kernel', name) else ('gemm_tn_kernel' if re.search(r'gemm_tn(2b?|b|256|16|16x256)?_kernel', name) else None)
    if not k:
        continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen[k]:
        seen[k].add(r['Dispatch_Id'])
        acc[k]['_ns'] += dur[r['Dispatch_Id']][0]
res = {}
for k, c in acc.items():
    ns = c['_ns']
    clk = c['GRBM_GUI_ACTIVE'] / 8.0 / ns            # GHz
    cycles = c['GRBM_GUI_ACTIVE'] / 8.0
    res[k] = {"launches": len(seen[k]), "effective_clock_GHz": clk,
              "mfma_busy_frac_at_that_clock": c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cycles),
              "wave_cycle_split": {n: c[n] / max(c['SQ_WAVE_CYCLES'], 1.0) for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY')},
              "note": "sums over the profiled launches; profiled passes clock a few % lower than un-profiled runs"}
json.dump(res, open(out_json, 'w'), indent=1)
print(json.dumps(res, indent=1))
