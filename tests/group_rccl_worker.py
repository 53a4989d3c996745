"""One rank of tests/test_distributed.py::test_group_abi_two_ranks_rccl: fh_group_* (RCCL ncclSend / ncclRecv behind the C ABI)
between two processes, one GPU each.  The 128-byte id travels over a gloo process group.
    python tests/group_rccl_worker.py <rank> <world> <port>"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = (int(x) for x in sys.argv[1:4])
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist

    import fenris_amd as fa
    from fenris_amd import _ffi

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank)
    lib = _ffi.lib()
    eng = fa.Engine(rank)
    idbuf = (C.c_uint8 * 128)()
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        assert lib.fh_group_unique_id(idbuf) == 0
        t.copy_(torch.tensor(list(idbuf), dtype=torch.uint8))
    dist.broadcast(t, 0)
    for i, b in enumerate(t.tolist()):
        idbuf[i] = b
    g = C.c_void_p()
    assert lib.fh_group_create(eng._h, idbuf, rank, world, C.byref(g)) == 0, eng.last_error()
    n = C.c_int(0)
    assert lib.fh_group_size(g, C.byref(n)) == 0 and n.value == world
    vals = torch.arange(1000, dtype=torch.float64, device=f"cuda:{rank}") * (rank + 1)
    # every rank but the first sends values[10:20] to the rank below, which adds them to its values[100:110]
    snd = (rank - 1, 10, 10) if rank > 0 else (-1, 0, 0)
    rcv = (rank + 1, 100, 10) if rank + 1 < world else (-1, 0, 0)
    assert lib.fh_group_set_exchange(g, snd[0], snd[1], snd[2], rcv[0], rcv[1], rcv[2]) == 0
    for rep in range(2):   # twice: the group is reusable
        assert lib.fh_group_exchange_start(g, C.c_void_p(vals.data_ptr())) == 0, eng.last_error()
        assert lib.fh_group_exchange_finish(g, C.c_void_p(vals.data_ptr())) == 0, eng.last_error()
    eng.synchronize()
    torch.cuda.synchronize()
    got = vals.cpu()
    exp = torch.arange(1000, dtype=torch.float64) * (rank + 1)
    if rank + 1 < world:
        exp[100:110] += 2 * torch.arange(10, 20, dtype=torch.float64) * (rank + 2)
    assert torch.equal(got, exp), (got[95:115], exp[95:115])
    dist.barrier()
    lib.fh_group_destroy(g)
    eng.close()
    dist.destroy_process_group()
    print(f"rank {rank} ok rccl_ranks={n.value}", flush=True)


if __name__ == "__main__":
    main()
