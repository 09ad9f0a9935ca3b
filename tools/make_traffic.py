"""HBM bytes per launch of one kernel family from the two rocprofv3 PMC summaries (tools/rocpd_pmc_stats.py output).

    python tools/make_traffic.py <pmc_fetch_size.json> <pmc_write_size.json> <out.json> <kernel substring>=<profile key> ...

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
counts the 128-byte requests of wide coalesced loads as 64 bytes -> doubled."""
import json
import sys


def family(path, sub):
    k = json.load(open(path))["kernels"]
    n = sum(v["dispatches"] for name, v in k.items() if sub in name)
    tot = sum(v["total"] for name, v in k.items() if sub in name)
    return n, tot


def main(fetch, write, out, *pairs):
    acc = {}
    for pair in pairs:   # several kernel-name substrings may feed one profile key (flat + halo-tile + split-K kernels)
        sub, key = pair.split("=")
        nf, tf = family(fetch, sub)
        nw, tw = family(write, sub)
        if not nf or not nw:
            continue
        a = acc.setdefault(key, {"subs": [], "nf": 0, "tf": 0.0, "nw": 0, "tw": 0.0})
        a["subs"].append(sub)
        a["nf"] += nf; a["tf"] += tf; a["nw"] += nw; a["tw"] += tw
    fams = {key: {"kernel": " + ".join(a["subs"]) + " (all tile variants)", "dispatches": a["nf"],
                  "fetch_size_kb_avg": a["tf"] / a["nf"], "write_size_kb_avg": a["tw"] / a["nw"],
                  "hbm_bytes_per_launch": (2.0 * a["tf"] / a["nf"] + a["tw"] / a["nw"]) * 1024.0}
            for key, a in acc.items()}
    res = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) over bench.py --steps 1 "
                  "--warmup 0; tools/profile_bench.sh",
        "gfx950_correction": "FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced loads "
                             "(MI355X_MICROARCH.md, HBM section): doubled",
        "families": fams,
    }
    open(out, "w").write(json.dumps(res, indent=1) + "\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
