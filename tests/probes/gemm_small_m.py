"""NT GEMM at the micro-batch sizes of the reference's default config (8192 token rows): eight-phase 256x256 kernel vs the
tiled 128x128 / 256x128 kernels (COMMU_GEMM8_OFF=1), per shape."""
import os, subprocess, sys, time
if len(sys.argv) > 1:
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
    import torch
    from commu_amd import ops
    def t(f, n=30):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    for M, N, K in [(8192, 1536, 512), (8192, 1024, 512), (8192, 512, 1024), (8192, 512, 512), (8192, 512, 1536), (73728, 1024, 512),
                    (65536, 512, 512), (65536, 1024, 512), (65536, 512, 1024), (32768, 1024, 512), (32768, 512, 1024), (32768, 1536, 512),
                    (16384, 1024, 1024), (16384, 1024, 2048), (16384, 2048, 1024), (16384, 3072, 1024), (49152, 512, 512), (24576, 1024, 512)]:
        A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        us = t(lambda: ops.gemm_nt(A, W, out=out))
        print(f"{M}x{N}x{K}: {us:7.1f} us {2*M*N*K/us/1e6:6.0f} TF")
else:
    for env in ({}, {"COMMU_GEMM8_ALWAYS": "1"}, {"COMMU_GEMM8_OFF": "1"}):
        print("env", env, flush=True)
        subprocess.run([sys.executable, __file__, "x"], env={**os.environ, **env})
