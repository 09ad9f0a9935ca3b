"""Time pp_window_attention alone at the BASELINE cfg-2 geometry on the MI355X (run via gpurun).

Token grid 30x54 (6x6 windows of 5x9), t frames per transformer call, nt = t // 2 key frames, 91 pooled keys; the masked
window set is the one the synthetic clip's dilated centre-third box produces (tools/bench_attention.py --masked N overrides).
Prints us per launch, algorithmic TFLOP/s (4 * nq * nk * 128 per masked (window, head); 4 * 45 * 45 * 128 per unmasked
(frame, window, head)) and the max error against a torch fp32 attention of two probe windows."""
import argparse
import math
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--t", type=int, default=18)
    ap.add_argument("--fh", type=int, default=30)
    ap.add_argument("--fw", type=int, default=54)
    ap.add_argument("--masked", type=int, default=-1, help="number of masked windows (default: centre block like cfg 2)")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    if os.environ.get("PP_LIB"):   # an instrumented build of the library (tools/trace_attention.sh)
        lib._lib = lib.Library(Path(os.environ["PP_LIB"]).resolve(), is_emulator=False)
    lib.load()
    dev = torch.device("cuda:0")
    t, fh, fw = args.t, args.fh, args.fw
    Hp, Wp = math.ceil(fh / 5) * 5, math.ceil(fw / 9) * 9
    nwh, nww = Hp // 5, Wp // 9
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(t, Hp, Wp, 1536, generator=g) * 0.7).half().to(dev)
    npool = (Hp // 4) * (Wp // 4)
    pkv = (torch.randn(t, npool, 1024, generator=g) * 0.7).half().to(dev)
    flags = torch.zeros(nwh, nww, dtype=torch.int32)
    if args.masked < 0:
        # cfg 2: box rows H/3..2H/3, cols W/3..2W/3, dilated -> token rows 9..20 of 30, cols 17..36 of 54 (approx.)
        flags[1:5, 1:5] = 1
    else:
        flags.view(-1)[: args.masked] = 1
    flags = flags.flatten().to(dev)
    nmask = int(flags.sum())
    t_ind = torch.arange(1, t, 2, dtype=torch.int32, device=dev)
    nt = t_ind.numel()
    out = torch.empty(t, fh, fw, 512, dtype=torch.float16, device=dev)
    for _ in range(3):
        ops.window_attention(qkv, pkv, flags, t_ind, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        ops.window_attention(qkv, pkv, flags, t_ind, out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.reps
    nk = nt * (45 + 148 + npool)
    flop = nmask * 4 * (4.0 * 45 * t * nk * 128) + (nwh * nww - nmask) * 4 * t * (4.0 * 45 * 45 * 128)
    print(f"window_attention t={t} grid {fh}x{fw} masked {nmask}/{nwh * nww} nk={nk}: {us:.1f} us/launch, "
          f"{flop / 1e9:.1f} GFLOP -> {flop / us / 1e6:.0f} TFLOP/s ({100 * flop / us / 1e6 / 2500:.1f} % of 2.5 PF)")
    # ---- spot check against torch fp32 on one masked and one unmasked window (head 1) ----------------------
    fl = flags.cpu().view(nwh, nww)
    head = 1
    q_all = qkv[..., 0:512].float()
    k_all = qkv[..., 512:1024].float()
    v_all = qkv[..., 1024:1536].float()
    hs = slice(head * 128, head * 128 + 128)
    worst = 0.0
    for want in (1, 0):
        idx = (fl == want).nonzero()
        if not len(idx):
            continue
        wi, wj = [int(v) for v in idx[len(idx) // 2]]
        ys, xs = slice(5 * wi, 5 * wi + 5), slice(9 * wj, 9 * wj + 9)
        if want:
            q = q_all[:, ys, xs, hs].reshape(-1, 128)
            ks, vs = [], []
            eh, ew = 3, 5
            rows = [r - eh for r in range(5)] + [r + eh for r in range(5)]
            cols = [c - ew for c in range(9)] + [c + ew for c in range(9)]
            for fr in t_ind.tolist():
                ks.append(k_all[fr, ys, xs, hs].reshape(-1, 128)); vs.append(v_all[fr, ys, xs, hs].reshape(-1, 128))
                for dr in rows:
                    for dc in cols:
                        if 0 <= dr < 5 and 0 <= dc < 9:
                            continue
                        y, x = (5 * wi + dr) % Hp, (9 * wj + dc) % Wp
                        ks.append(k_all[fr, y, x, hs][None]); vs.append(v_all[fr, y, x, hs][None])
                ks.append(pkv[fr, :, hs].float()); vs.append(pkv[fr, :, 512 + head * 128: 512 + head * 128 + 128].float())
            K, V = torch.cat(ks), torch.cat(vs)
            assert K.shape[0] == nk
            ref = torch.softmax(q @ K.t() / math.sqrt(128), -1) @ V
            ref = ref.view(t, 5, 9, 128)
        else:
            q = q_all[:, ys, xs, hs].reshape(t, 45, 128)
            K = k_all[:, ys, xs, hs].reshape(t, 45, 128)
            V = v_all[:, ys, xs, hs].reshape(t, 45, 128)
            ref = (torch.softmax(q @ K.transpose(1, 2) / math.sqrt(128), -1) @ V).view(t, 5, 9, 128)
        got = out[:, :, :, hs].float()
        y1, x1 = min(5 * wi + 5, fh), min(9 * wj + 9, fw)
        err = (got[:, 5 * wi:y1, 9 * wj:x1] - ref[:, : y1 - 5 * wi, : x1 - 9 * wj]).abs().max().item()
        worst = max(worst, err)
        print(f"  window ({wi},{wj}) {'masked' if want else 'unmasked'}: max abs err {err:.2e} (ref absmax {ref.abs().max().item():.2f})")
    assert worst < 5e-3, worst


if __name__ == "__main__":
    main()
