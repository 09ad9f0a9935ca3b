"""graphs.GraphCache (hipGraph capture of fixed-shape sweeps): a captured sweep must behave like the eager function from its FIRST
call on -- also when the sweep writes its own inputs (r06: the transformer blocks update their token tensor in place; the first
replay of a process ran on the warm-up run's output until the static inputs were refilled after the capture)."""
import pytest
import torch

from comfyui_propainter_nodes_amd import graphs


@pytest.mark.gpu
def test_graph_cache_first_call_and_replays_match_eager(hip_lib):
    dev = torch.device("cuda:0")

    def sweep(x, y):          # writes its first input in place, like InpaintGeneratorMI355._transformer
        x.mul_(2.0)
        x.add_(y)
        return x * 3.0

    cache = graphs.GraphCache()
    for i in range(3):
        x = torch.full((5, 7), float(i + 1), device=dev)
        y = torch.full((5, 7), 0.5, device=dev)
        got = cache.run(("sweep",), sweep, x.clone(), y)
        assert torch.equal(got, (x * 2.0 + y) * 3.0), i
    # the gathered-input form: `fill` writes the static inputs itself
    cache2 = graphs.GraphCache(max_entries=1)
    src = torch.arange(24, device=dev, dtype=torch.float32).view(6, 4)
    for i in range(3):
        rows = torch.tensor([i, i + 2], device=dev)

        def fill(bufs):
            if bufs is None:
                return [src.index_select(0, rows)]
            torch.index_select(src, 0, rows, out=bufs[0])
            return bufs

        got = cache2.run_filled(("gathered",), lambda a: (a.mul_(2.0), a + 1.0)[1], fill, dev)
        assert torch.equal(got, src[[i, i + 2]] * 2.0 + 1.0), i
