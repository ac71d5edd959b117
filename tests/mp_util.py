"""Spawns `world` gloo ranks on 127.0.0.1 running fn(rank, world); any exception fails the test."""
import os
import socket
import traceback

import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _entry(rank, world, port, fn, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        import torch.distributed as dist
        from nerfactor_amd import dist as nd
        nd.init_from_env(backend='gloo')
        fn(rank, world)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, None))
    except Exception:  # noqa: BLE001 - reported to the parent
        q.put((rank, traceback.format_exc()))


def run_workers(fn, world=2, timeout=180):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_entry, args=(r, world, port, fn, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(60)
    errors = [e for _, e in results if e]
    assert not errors, '\n'.join(errors)
