import subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import torch
    from mmf_amd import _native as nat
    cfg, M, N, K = map(int, sys.argv[1:5])
    nat.set_tunable(nat.TUN_GEMM_WIDE, cfg)
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); C0 = torch.empty_like(C)
    nat.gemm(A, B, C, M, N, K, K, K, N)
    torch.cuda.synchronize()
    nat.gemm(A, B, C0, M, N, K, K, K, N, debug_flags=(1 << 17) | (1 << 13))
    torch.cuda.synchronize()
    print("equal" if torch.equal(C, C0) else "DIFFERENT max %g" % float((C.float() - C0.float()).abs().max()))
    sys.exit(0)
for cfg, N in ((1, 768), (1, 192), (2, 2304), (2, 384), (3, 3072)):
    for M, K in ((7296, 768), (512, 128), (512, 192), (512, 256), (1500, 768), (3200, 3072)):
        r = subprocess.run([sys.executable, __file__, str(cfg), str(M), str(N), str(K)], capture_output=True, text=True, timeout=120)
        print(cfg, M, N, K, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1], flush=True)
