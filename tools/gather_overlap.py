"""Per-step cost of the hit exchange on rank 0: no exchange / blocking gather / pipelined start+finish /
pipelined with the step on the library's helper thread ("async", what bench.py does with several ranks).

Runs ONE rank with backend "nccl" (RCCL) on cuda:0, so the collective itself is a local copy; what is
measured is what the exchange adds to rank 0's step loop (staging copies, collective launch, host
synchronisation) and how much of it the pipelined form hides behind the next scan.

usage: python tools/gather_overlap.py [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]


def main():
    import torch
    import torch.distributed as dist
    from kmergma_amd import _lib, parallel, workloads
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    refs = workloads.fixture_refs(os.path.join(ROOT, "tests", "data"), 6)
    ctx = _lib.Context(0)
    ctx.set_refs(6, [refs["RV"]], [refs["ws"]], [30.0], [refs["N"]])
    genome, _ = workloads.make_chr22_like(ctx, refs["genes"], seed=22)
    g = parallel.HitGatherer(device=dev, capacity=256)

    def scan():
        return ctx.step_hits(genome, _lib.MODE_SINGLE, 50, 0, 0)

    def run(mode):
        pending = None
        if mode == "async":                      # bench.py's loop for several ranks: kgma_step_begin / kgma_step_end
            ctx.step_begin(genome, _lib.MODE_SINGLE, 50, 0, 0)
            for it in range(steps + 10):
                if it == 10:
                    t0 = time.perf_counter()
                hits = ctx.step_end()
                if it + 1 < steps + 10:
                    ctx.step_begin(genome, _lib.MODE_SINGLE, 50, 0, 0)
                slot = g.start(hits, 0, 0)
                if pending is not None:
                    g.finish(pending)
                pending = slot
            g.finish(pending)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / steps
        for it in range(steps + 10):
            if it == 10:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            hits = scan()
            if mode == "blocking":
                g.gather(hits, 0, 0)
            elif mode == "pipelined":
                slot = g.start(hits, 0, 0)
                if pending is not None:
                    g.finish(pending)
                pending = slot
        if pending is not None:
            g.finish(pending)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / steps

    for reserved in (0, 8):                       # kgma_set_reserved_cus: CUs left free for the collective
        ctx.set_reserved_cus(reserved)
        for mode in ("none", "blocking", "pipelined", "async", "none", "async"):
            print(f"reserved CUs {reserved}: {mode:10s} {run(mode):.4f} ms/step", flush=True)
    genome.free()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
