"""Where do the pixels beyond 2 LSB come from?  (cfg3_80f_node / cfg5_160f_node, fp16 "enable", max 3 LSB.)

Two legs:

  --run   (MI355X)  run a node fixture through OUR node method and, for every masked byte that ends more than `--lsb` from the
          reference's, dump where it is (frame, y, x, channel), both bytes, and OUR generator output (x255, before any truncation)
          of every window that has the frame among its local frames -> gpurun_out/outliers_<case>_<fp16>.npz
  --explain (build container)  join that dump with the reference's own float pred_img of the same windows
          ($TMPDIR/raw_predimg_<case>.npz, written by tests/golden/make_predimg.py) and replay the reference's compose chain
          (propainter_inference.py:283-307: astype(uint8) per window, 0.5 / 0.5 blend, astype(uint8)) on both sides: prints per
          pixel the float difference per window and whether the byte difference is the truncation cascade of sub-LSB float
          differences or a float outlier.
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def compose_chain(vals255):
    """The reference's per-pixel compose over the windows that visit a frame, in window order (:296-307)."""
    c = None
    for v in vals255:
        b = np.uint8(np.float32(v))                       # np.array(pred_img[i]).astype(np.uint8): truncation
        c = b if c is None else np.uint8(np.float32(c) * np.float32(0.5) + np.float32(b) * np.float32(0.5))
    return int(c)


def leg_run(case: str, fp16: str, lsb: int) -> None:
    os.environ.setdefault("PP_ALLOW_SYNTHETIC_WEIGHTS", "1")
    import torch  # noqa: F401

    import node_case
    from comfyui_propainter_nodes_amd import pipeline

    det: dict = {}
    m = node_case.evaluate_node_case(case, fp16, detail=det)
    P, g = det["params"], det["fixture"]
    md, out_u8 = det["md"].astype(bool), det["out_u8"]
    T, h, w = md.shape
    okeep = g["out_keep"] if "out_keep" in g.files else np.arange(T)
    sel = md.copy()
    drop = np.ones(T, bool)
    drop[okeep] = False
    sel[drop] = False
    want = np.zeros_like(out_u8)
    want[sel] = g["out_masked"]
    diff = np.abs(out_u8.astype(np.int32) - want.astype(np.int32)) * sel[..., None]
    pos = np.argwhere(diff > lsb)                            # [n, 4] = (t, y, x, c)
    sched = pipeline.window_schedule(pipeline.ProPainterConfig(P["ref_stride"], P["neighbor_length"], P["subvideo_length"],
                                                               P["raft_iter"], fp16, T, torch.device("cpu"), (w, h)))
    # the completed flows at the fixture's sub-grid (stride s, f16): how far OUR flows are from the reference's around a pixel
    s_ = P["flow_stride"]
    fkeep = list(g["flow_keep"]) if "flow_keep" in g.files else list(range(T - 1))
    ours_pf = det["trace"]["pred_flows"].cpu()[:, :, ::s_, ::s_].permute(0, 1, 4, 2, 3).numpy()     # [2,T-1,2,h/s,w/s]
    d_pf = np.abs(ours_pf[:, fkeep] - g["pred_flow"].astype(np.float32)).max(axis=2)                 # [2,nkeep,h/s,w/s]
    print(f"completed-flow difference to the reference over the whole clip (stored sub-grid): max {d_pf.max():.2f} px, "
          f"{(d_pf > 0.5).mean():.2e} of the grid points beyond 0.5 px")

    def flow_diff_near(t, y, x):
        out = []
        for ft in (t - 1, t):
            if ft in fkeep:
                k = fkeep.index(ft)
                yy, xx = min(d_pf.shape[2] - 1, int(round(y / s_))), min(d_pf.shape[3] - 1, int(round(x / s_)))
                y0, y1, x0, x1 = max(0, yy - 1), yy + 2, max(0, xx - 1), xx + 2
                out.append(float(d_pf[:, k, y0:y1, x0:x1].max()))
        return max(out) if out else float("nan")

    rows = []
    for t, y, x, c in pos:
        visits = [(wi, nb.index(int(t))) for wi, (nb, _) in enumerate(sched) if int(t) in nb]
        ours = [float((det["trace"]["pred_imgs"][wi][i][y, x, c] + 1) / 2 * 255) for wi, i in visits]
        # index of (y, x) among the masked pixels of frame t (the order make_predimg.py stores them in)
        k = int(md[t].ravel()[: y * w + x].sum())
        rows.append((t, y, x, c, int(out_u8[t, y, x, c]), int(want[t, y, x, c]), k, visits, ours))
        print(f"frame {t} ({y},{x}) ch {c}: ours {out_u8[t, y, x, c]} reference {want[t, y, x, c]}; windows "
              + ", ".join(f"w{wi}[{i}] {v:.3f}" for (wi, i), v in zip(visits, ours)) + f" -> replay {compose_chain(ours)}; "
              f"completed flows within 1 grid step of the pixel (frames {t - 1}, {t}) differ from the reference's by up to {flow_diff_near(t, y, x):.2f} px")
    out = ROOT / "gpurun_out" / f"outliers_{case}_{fp16}.npz"
    out.parent.mkdir(exist_ok=True)
    np.savez(out, pos=pos, k=np.array([r[6] for r in rows], dtype=np.int64), ours_u8=np.array([r[4] for r in rows]),
             ref_u8=np.array([r[5] for r in rows]), visits=np.array([str(r[7]) for r in rows]),
             ours_255=np.array([str(r[8]) for r in rows]), metrics=np.array(str(m)))
    print(f"{len(rows)} bytes beyond {lsb} LSB of {int(sel.sum()) * 3} compared -> {out}")


def leg_explain(case: str, fp16: str) -> None:
    d = np.load(ROOT / "gpurun_out" / f"outliers_{case}_{fp16}.npz")
    raw = np.load(Path(tempfile.gettempdir()) / f"raw_predimg_{case}.npz")
    n_cascade = 0
    for (t, y, x, c), k, ou, ru, visits, ours in zip(d["pos"], d["k"], d["ours_u8"], d["ref_u8"], d["visits"], d["ours_255"]):
        visits, ours = eval(str(visits)), eval(str(ours))
        ref = [float((raw[f"w{wi}_f{i}"][k, c] + 1) / 2 * 255) for wi, i in visits]
        assert compose_chain(ref) == ru, (compose_chain(ref), ru)       # the capture reproduces the reference's byte
        dfl = [abs(a - b) for a, b in zip(ours, ref)]
        cascade = max(dfl) < 2.0
        n_cascade += cascade
        print(f"frame {t} ({y},{x}) ch {c}: bytes {ou} vs {ru}; x255 floats ours {['%.3f' % v for v in ours]} reference "
              f"{['%.3f' % v for v in ref]}; max float diff {max(dfl):.3f} LSB = {max(dfl) / 255:.2e} in pixel units -> "
              + ("truncation cascade of sub-2-LSB float differences" if cascade else "FLOAT OUTLIER"))
    print(f"{n_cascade} of {len(d['pos'])} explained as truncation cascades")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True)
    ap.add_argument("--fp16", default="enable")
    ap.add_argument("--lsb", type=int, default=2)
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--explain", action="store_true")
    a = ap.parse_args()
    if a.run:
        leg_run(a.case, a.fp16, a.lsb)
    if a.explain:
        leg_explain(a.case, a.fp16)
