"""One rank of tests/test_distributed.py::test_overlapped_slab_assembly_two_processes_one_device: a fresh process (started
before anything touches the GPU) that runs fenris_amd.distributed.SlabAssembly with overlap under a gloo process group, all
ranks on cuda:0, and checks its owned rows against the oracle's single-mesh matrix.
    python tests/slab_overlap_worker.py <rank> <world> <port> <cells> <layers_total>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, cells, units_z = (int(x) for x in sys.argv[1:6])
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import numpy as np
    import torch
    import torch.distributed as dist

    import fenris_amd as fa
    from fenris_amd import distributed as fd
    from fenris_amd import quadrature
    from oracle import oracle
    from test_distributed import LAME, _check_owned_rows, _global_reference

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    slab = fd.make_slab(1.0, 1, 1, units_z, cells, rank, world)
    w, p = quadrature.tensor.hexahedron_gauss(2)
    lame = fa.LameParameters(*LAME)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
    op = fa.MaterialEllipticOperator(fa.LinearElasticMaterial())

    def configure(engine, mesh):
        return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(op)
                .with_quadrature_table(qt).with_u(None).build())

    asm = fd.SlabAssembly(slab, configure, device=0, overlap=True, stream=torch.cuda.current_stream().cuda_stream)
    assert (asm.split is not None) == (rank > 0)      # ranks with a ghost plane below launch its rows first
    assert asm.comm is not None
    for _ in range(2):                                 # twice: overwrite semantics, buffers reused
        asm.enqueue(fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    asm.poll_status()
    torch.cuda.synchronize()
    dist.barrier()
    ro, ci = asm.main.pattern()
    gro, gci, gvals = _global_reference(oracle, units_z, cells, oracle.LINEAR_ELASTIC)
    _check_owned_rows(slab, 3, ro, ci, asm.values.cpu().numpy(), gro, gci, gvals)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok kernel={asm.main.last_kernel_name()}", flush=True)


if __name__ == "__main__":
    main()
