"""Weight-gradient GEMM (split launch + deterministic reduction) on the small reduced-row counts of the 512-ray / stage-2 batches:
time by split count S, to check the rule of nu_wgrad_pick_split (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import tn, lib

if __name__ == "__main__":
    for P in (7168, 14256, 33024, 67456):
        for (n1, n2, pairs) in ((256, 256, 1), (256, 256, 2), (1024, 288, 1), (256, 64, 1), (128, 288, 1)):
            rule = lib.nu_wgrad_pick_split(P, n1, n2, 1, 0)
            print(f"-- P={P} {n1}x{n2} pairs={pairs}: rule picks S={rule}")
            for S in sorted(set([max(rule // 4, 1), max(rule // 2, 1), rule, rule * 2, rule * 4])):
                if S * 32 > P:
                    continue
                tn(P, n1, n2, pairs, S)
