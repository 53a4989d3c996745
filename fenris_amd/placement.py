"""Set-up helpers around fh_time_assembly_dev / fh_tune_placement_dev: bring a fresh device to its steady rate, then keep the
better of several allocations of the values array and of the library's record buffer.  Used by bench.py and by SlabAssembly (one rank = one
GPU: rank-local, no collective); everything here happens BEFORE a timed region."""
import time

MAX_LIVE_BYTES = 160e9   # candidates stay allocated while the probe runs (a new one gets other memory): keep them below this
FREE_RESERVE = 0.25      # ... and below this share of what the device reports free (the library's record candidates come on top)


def settle_device(eng, values, flags, max_s=2.5, group=20):
    """Untimed assemblies until their time stops improving (or max_s seconds): a fresh box reaches its steady rate only after about a
    second of work (profiles/r03_affine_experiments.txt 12).  Returns {groups, assemblies_per_group, first_ms, last_ms, seconds}."""
    t0 = time.perf_counter()
    seen, best, flat = [], float("inf"), 0
    while time.perf_counter() - t0 < max_s:
        ms = eng.time_assembly(values, flags, group)
        seen.append(ms)
        flat = flat + 1 if ms > 0.997 * best else 0
        best = min(best, ms)
        if flat >= 3 and len(seen) >= 4:
            break
    return {"groups": len(seen), "assemblies_per_group": group, "first_ms": round(seen[0], 4), "last_ms": round(seen[-1], 4),
            "seconds": round(time.perf_counter() - t0, 3)}


def probe_placement(eng, values, flags, tries):
    """The better of several allocations of the values array (and, inside the library, of the element records): the time of the
    owner-computes kernels follows how these buffers happen to be backed by device memory -- the same context and arguments run at one
    of several levels up to 10 % apart for the life of an allocation, and often only one allocation in five is at the fast level.  Every
    trial is three real assemblies.  Returns (values, report)."""
    import torch

    nbytes = max(values.numel() * values.element_size(), 1)
    budget = MAX_LIVE_BYTES
    if values.is_cuda:   # a smaller or shared device: what is free now bounds the candidates, not the 288 GB of an idle MI355X
        free, _ = torch.cuda.mem_get_info(values.device)
        budget = min(budget, (1.0 - FREE_RESERVE) * free + nbytes)
    tries = max(0, min(int(tries), int(budget // nbytes) - 1))
    seen = [eng.time_assembly(values, flags)]
    best, rejected = values, []
    for k in range(tries):
        try:
            cand = torch.zeros_like(values)      # the previous candidates stay allocated: a new one gets other memory
        except torch.OutOfMemoryError:           # keep the best so far
            tries = k
            break
        t = eng.time_assembly(cand, flags)
        if t < 0.98 * min(seen):
            rejected.append(best)
            best = cand
        else:
            rejected.append(cand)
        seen.append(t)
    before, after = eng.tune_placement(best, flags, min(tries, 3)) if tries > 0 else (seen[0], seen[0])
    del rejected
    torch.cuda.empty_cache()
    return best, {"values_ms_seen": [round(x, 4) for x in seen], "records_ms_before": round(before, 4), "records_ms_after": round(after, 4),
                  "tries": tries}
