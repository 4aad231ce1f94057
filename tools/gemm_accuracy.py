"""Error of each GEMM arithmetic mode against an fp64 product on the model's shapes (run on the GPU box).

Prints rms and max error relative to the rms of the exact result, for: mode 0 (fp32 MFMA), mode 1 (bf16x3 split), mode 2
(bf16), and - as the yardstick of what "fp32" means on the reference's side - torch's CPU fp32 matmul."""
import importlib, json, os, sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("dense2sparse-vit_amd.d2s.ops")


def main():
    dev = torch.device("cuda:0")
    out = []
    for (M, N, K, scale) in [(4096, 1536, 384, 1.0), (4096, 384, 1536, 1.0), (4096, 1152, 384, 30.0), (4096, 384, 384, 1e-3)]:
        g = torch.Generator().manual_seed(M + N + K)
        x = torch.randn(M, K, generator=g) * scale
        w = torch.randn(N, K, generator=g) * 0.02
        ref = x.double() @ w.double().t()
        rms = ref.pow(2).mean().sqrt().item()
        row = {"M": M, "N": N, "K": K, "x_scale": scale}
        cpu = (x @ w.t()).double()
        row["torch_cpu_fp32"] = {"rms": ((cpu - ref).pow(2).mean().sqrt() / rms).item(), "max": ((cpu - ref).abs().max() / rms).item()}
        for mode, name in [(0, "exact_fp32_mfma"), (1, "bf16x3_split"), (2, "bf16")]:
            ops.set_gemm_mode(mode)
            y = ops.linear_fwd(x.to(dev), w.to(dev)).cpu().double()
            row[name] = {"rms": ((y - ref).pow(2).mean().sqrt() / rms).item(), "max": ((y - ref).abs().max() / rms).item()}
        ops.set_gemm_mode(0)
        out.append(row)
        print(json.dumps(row))
    return out


if __name__ == "__main__":
    main()
