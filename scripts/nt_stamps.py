"""Development aid: per-workgroup start / end stamps of one fp32 NT launch (needs the stamp hooks compiled in)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nu_nerf_amd import _lib as L
from nu_nerf_amd.engine import GemmNT, addr
lib = L.load(); dev = torch.device('cuda:0')

def run(M, N, K, epi):
    A = torch.randn(M, K, device=dev); B = torch.randn((N + 127) // 128 * 128, K, device=dev) / K ** 0.5
    C = torch.empty(M, N, device=dev); C2 = torch.empty(M, N, device=dev)
    H = torch.rand(M, N, device=dev); D = torch.randn(M, N, device=dev); b = torch.randn(N, device=dev)
    g = GemmNT(addr(A), K, addr(B), K, M, N, K, addr(C), N, addr(C2), N, addr(b), addr(H), N, addr(D), N, addr(D), N,
               0, 0, 1.0, 1, 0, 0, 0, 0, 0, 0, 0, 0, epi)
    for _ in range(5):
        L.check(lib.nu_gemm_nt_ex(ctypes.byref(g), L.stream()), "nt")
    torch.cuda.synchronize()
    out = np.zeros(36 * 1024, dtype=np.uint64)
    assert lib.nu_debug_nt2_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
    np.save(os.path.join(ROOT, 'gpurun_out', f'nt_stamps_{M}_{K}_{epi}.npy'), out)
    s = out[:4096].reshape(1024, 4)[:512]
    t0 = s[:, 0].astype(np.int64); t1 = s[:, 1].astype(np.int64); xcc = (s[:, 2] >> np.uint64(32)).astype(np.int64); hw = (s[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    base = t0.min(); us = 0.01     # 100 MHz
    st = (t0 - base) * us; en = (t1 - base) * us; dur = en - st
    print(f"M={M} N={N} K={K} epi={epi}: tiles/WG {s[:,3].min()}..{s[:,3].max()}")
    print(f"  start  min {st.min():.1f} p50 {np.median(st):.1f} max {st.max():.1f} us")
    print(f"  end    min {en.min():.1f} p10 {np.percentile(en,10):.1f} p50 {np.median(en):.1f} p90 {np.percentile(en,90):.1f} max {en.max():.1f} us")
    print(f"  dur    min {dur.min():.1f} p50 {np.median(dur):.1f} max {dur.max():.1f} us")
    for x in range(8):
        m = (xcc & 15) == x
        if m.any(): print(f"    xcc {x}: n {m.sum():3d} end mean {en[m].mean():.1f} min {en[m].min():.1f} max {en[m].max():.1f}; start mean {st[m].mean():.1f}")
    cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1
    key = (xcc & 15) * 1000 + se * 100 + sh * 50 + cu
    u, cnt = np.unique(key, return_counts=True)
    print(f"  distinct (xcc,se,sh,cu) {len(u)}; workgroups per CU histogram {np.bincount(cnt)}")
    # pairs sharing a CU: are they blockIdx i and i + 256 ?
    same = sum(1 for i in range(256) if key[i] == key[i + 256])
    print(f"  blockIdx i and i+256 on the same CU: {same} of 256;   i and i+1: {sum(1 for i in range(0,512,2) if key[i]==key[i+1])} of 256;  i and i+8: {sum(1 for i in range(504) if key[i]==key[i+8])}")

for a in ((262144, 256, 256, 1), (524288, 256, 256, 1), (262144, 256, 1024, 7), (262144, 256, 256, 3)):
    run(*a)
