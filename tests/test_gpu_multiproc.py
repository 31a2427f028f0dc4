"""Row-sharded commit + prove with REAL process groups on the GPU box: `world` processes, one shard context each.
With fewer GPUs than ranks (the test box has one) all ranks share GPU 0 and gloo carries the exchange, since RCCL refuses
duplicate devices: lcpc_amd.distributed.sharded_commit / sharded_prove through torch.distributed.  With >= `world`
GPUs visible every rank takes its own GPU, the process group is "nccl" and the commit + prove additionally run through
the library's native RCCL exchange (lcpc_comm_init / lcpc_commit_sharded_device / lcpc_prove_sharded_rccl).  Either
way every rank must end with the oracle's root and the oracle's proof bytes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fid, n_rows, n_per_row, n_cols, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here, os.path.join(os.path.dirname(here), "oracle")]
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    real = torch.cuda.device_count() >= world          # one GPU per rank: RCCL; else all ranks on GPU 0 over gloo
    dev = rank if real else 0
    torch.cuda.set_device(dev)
    if real:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as O
        from common import mk_transcript
        from lcpc_amd import LigeroEncoding, Transcript
        from lcpc_amd.distributed import HipShardEngine, sharded_commit, sharded_prove
        enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, device=dev, shard=(rank, world))
        eng = HipShardEngine(enc)
        rb, re, _, _, _ = eng.layout(n_rows)
        coeffs = O.random_elems(fid, n_rows * n_per_row, 41).reshape(n_rows, n_per_row, -1)
        local = torch.from_numpy(coeffs[rb:re].copy().view(np.int64)).cuda()
        root = sharded_commit(eng, local, n_rows)
        outer = O.random_elems(fid, n_rows, 43)
        proof, cols = sharded_prove(eng, outer, mk_transcript(Transcript, root, enc.get_n_col_opens()))
        if real:      # the same through the library's own communicator
            eng.comm_init()
            root2 = eng.commit_native(local, n_rows)
            proof2, _ = eng.prove_native(outer, mk_transcript(Transcript, root2, enc.get_n_col_opens()))
            if root2 != root or proof2 != proof:
                raise RuntimeError("native RCCL exchange disagrees with the torch.distributed exchange")
            # LCPC_COMMIT_ASYNC_TAIL on two commitments of the encoder, filled alternately without host synchronisation
            # (what bench.py --gpus N times): their collectives share the communicator and must keep their order on every rank
            eng2 = HipShardEngine(enc)
            for _ in range(3):
                eng.commit_native(local, n_rows, want_root=False, async_tail=True)
                eng2.commit_native(local, n_rows, want_root=False, async_tail=True)
            if eng.cm.get_root() != root or eng2.cm.get_root() != root:
                raise RuntimeError("async-tail commits disagree with the sequential one")
            proof3, _ = eng2.prove_native(outer, mk_transcript(Transcript, root, enc.get_n_col_opens()))
            if proof3 != proof:
                raise RuntimeError("prove on an async-tail commitment disagrees")
        q.put((rank, root, proof))
    except Exception as e:      # surface the failure instead of a queue timeout
        q.put((rank, repr(e), b""))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fid,n_rows,n_per_row,n_cols", [(2, 3, 130, 128, 256), (3, 3, 70, 64, 128)])
def test_sharded_commit_and_prove_processes(oracle, world, fid, n_rows, n_per_row, n_cols):
    from common import mk_transcript
    O = oracle
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fid, n_rows, n_per_row, n_cols, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols)
    coeffs = O.random_elems(fid, n_rows * n_per_row, 41)
    oc = O.Commit.commit(coeffs, oenc)
    root = oc.get_root()
    outer = O.random_elems(fid, n_rows, 43)
    opf, _ = oc.prove(outer, oenc, mk_transcript(O.Transcript, root, oenc.get_n_col_opens()))
    for rank, r, proof in res:
        assert r == root, (rank, r)
        assert proof == opf, rank
