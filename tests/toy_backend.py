"""A toy stage backend for testing the multi-GPU orchestration on CPU (NOT a model: cheap deterministic
functions with the same data dependencies -- temporal coupling inside a chunk, halos, window locality,
order-dependent compose -- so that any mistake in sharding / halo / exchange logic changes the result)."""
import torch


class ToyBackend:
    def to_frames(self, frames_u8):
        return frames_u8.float() / 255.0 * 2 - 1

    def enc_tail(self, hw):
        return (1, 1, 4)

    def pred_tail(self, hw):
        return (hw[0], hw[1], 4)

    def raft(self, frames):
        a, b = frames[:-1, ..., :2], frames[1:, ..., :2]
        return torch.stack([b - a + 0.1 * a * b, a - b + 0.05 * a], 0)

    def complete(self, flows, masks):
        m = masks[:-1].float()[None, ..., None]
        t = torch.arange(flows.shape[1], dtype=torch.float32).view(1, -1, 1, 1, 1)
        # temporal coupling across the whole chunk (cumsum) + position inside the chunk: chunk-boundary sensitive
        return flows * (1 - m) + m * (0.3 * torch.cumsum(flows, 1) + 0.01 * t) + 0.001 * torch.flip(torch.cumsum(torch.flip(flows, [1]), 1), [1])

    def img_prop(self, frames, masks, flows):
        n = frames.shape[0]
        f = torch.cat([flows[0], flows[0][-1:] * 0], 0)[..., :1] if n > 1 else frames[..., :1] * 0
        back = torch.flip(torch.cumsum(torch.flip(frames, [0]), 0), [0])
        prop = 0.5 * frames + 0.05 * torch.cumsum(frames, 0) + 0.02 * back + 0.1 * f
        upd = (masks.bool() & (torch.cumsum(masks.long(), 0) % 2 == 1)).to(torch.uint8)
        return prop, upd

    def encode(self, frames, prop, md, upd):
        m = md.float()[..., None]
        x = frames * (1 - m) + prop * m
        return torch.stack([x.mean((1, 2, 3)), x.amax((1, 2, 3)), md.float().mean((1, 2)), upd.float().mean((1, 2))], 1).view(-1, 1, 1, 4).half()

    def make_state(self, enc, flows, md, upd):
        return {"enc": enc.float(), "flows": flows, "md": md, "upd": upd}

    def enc_landed(self, st, enc):
        """The halo rows of the encoder features arrived (run_rank posts that exchange and starts the feature propagation of its
        interior windows under it): this state holds a converted COPY, so it is refreshed; rows that were read too early would
        have been NaN (PP_POISON_HALOS=1 in the tests)."""
        st["enc"] = enc.float()

    def propagate_windows(self, st, windows):
        out = []
        for nb in windows:
            e = st["enc"][nb[0]:nb[-1] + 1]
            fl = st["flows"][:, nb[0]:nb[-1]].mean((2, 3, 4)) if len(nb) > 1 else torch.zeros(2, 0)
            # means, not sums: the output must stay sensitive to every flow / feature whatever the window length
            n = e.shape[0]
            w = torch.arange(1, n + 1, dtype=torch.float32).view(-1, 1, 1, 1)
            acc = torch.cumsum(e, 0) / w + torch.flip(torch.cumsum(torch.flip(e, [0]), 0) / w, [0])
            acc[1:] += 3.0 * fl[0].view(-1, 1, 1, 1)
            acc[:-1] += 3.0 * fl[1].view(-1, 1, 1, 1)
            out.append(acc)
        return out

    def forward_window(self, st, nb, refs, local_prop):
        H, W = st["md"].shape[1:]
        ref = st["enc"][refs].sum() if refs else torch.tensor(0.0)
        # (nothing here may depend on absolute frame numbers: a rank works on a clip state over its own frames + halos)
        # un-saturated on purpose (|val| stays well below 1): zero or wrong flows / features change the uint8 result
        val = torch.tanh(local_prop.mean((1, 2, 3)) * 0.25 + 0.002 * ref / max(len(refs), 1) + 0.001 * torch.arange(len(nb), dtype=torch.float32))
        ramp = torch.linspace(-0.2, 0.2, H * W).view(1, H, W, 1)
        return (val.view(-1, 1, 1, 1) * 0.7 + ramp).expand(-1, H, W, 4).contiguous().half()

    def compose(self, comp, pred, frame_ids, first, md, frames_u8):   # ids index comp / md / frames_u8 alike
        for j, idx in enumerate(frame_ids):
            p = ((pred[j, ..., :3].float() + 1) / 2 * 255).to(torch.uint8)
            m = md[idx].bool()[..., None]
            img = torch.where(m, p, frames_u8[idx])
            if first[j]:
                comp[idx] = img
            else:
                comp[idx] = (comp[idx].float() * 0.5 + img.float() * 0.5).to(torch.uint8)
