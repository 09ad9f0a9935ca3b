"""r06 diagnostic: are RAFT's flows the same bits whether the two pair-directions of a chunk run as one batch (PP_RAFT_LANES=1) or
as two half-batches on two streams (2)?  Sizes chosen so that the chunked eager path of cfg 5 (18 pairs per chunk at 1280x720) is hit."""
import os
import sys

import torch

sys.path.insert(0, '.')
from comfyui_propainter_nodes_amd import raft, weights, synth  # noqa: E402

dev = torch.device('cuda:0')
for (T, H, W, mvb, iters) in ((37, 720, 1280, 48 << 30, 2), (7, 720, 1280, 6 << 30, 4), (9, 360, 640, 48 << 30, 4)):
    image, _ = synth.synthetic_clip(T, H, W)
    frames = (image * 2 - 1).to(dev).contiguous()
    R = raft.RaftFlow(weights.synth_state_dicts(0)['raft'], dev, max_volume_bytes=mvb)
    os.environ['PP_RAFT_LANES'] = '1'
    a = R.bidirectional(frames, iters).clone()
    os.environ['PP_RAFT_LANES'] = '2'
    b = R.bidirectional(frames, iters)
    d = (a - b).abs()
    print(T, H, W, mvb >> 30, 'GB: lanes 2 == lanes 1:', bool(torch.equal(a, b)), 'max', float(d.max()), 'differing values', int((d > 0).sum()), flush=True)
    del R, a, b
    torch.cuda.empty_cache()
