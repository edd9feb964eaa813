"""nu_nerf_amd -- MI355X-native (gfx950) hot path of NU-NeRF's stage-1/2 training step.

Hand-written HIP kernels behind a C ABI (include/nu_nerf.h, nu_nerf_amd/csrc) surfaced as
torch.autograd.Function ops; the nn.Module boundary mirrors the reference's
`name2renderer[cfg['network']](cfg)` protocol (network/renderer_zerothick.py:2057-2060).
"""
__version__ = "0.1.0"
