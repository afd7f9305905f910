#!/usr/bin/env python3
"""Start one shared-mode worker per GPU of this node and (optionally) register them with a running front server.

  python scripts/serve_workers.py                       one worker per visible GPU on ports 5003, 5004, ...
  python scripts/serve_workers.py --server http://127.0.0.1:8000 --server-nonce <nonce>
                                                        ... and enter each into the server's executor table (POST /register)
  HIP_VISIBLE_DEVICES=0,1,2,3 python scripts/serve_workers.py --base-port 6000

The reference starts ONE worker (server/main.py:244-276, `python -m manga_translator shared` on port + 1) and leaves the GPU choice to
the framework; this starts the same worker protocol (manga_translator/mode/share.py:47-174) once per GPU, each process pinned with
HIP_VISIBLE_DEVICES=<i> before it imports torch, with the HIP plugins as its detector / ocr / inpainter
(manga_image_translator_amd/serve.py).  Ctrl-C stops the workers."""
import argparse
import os
import signal
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from manga_image_translator_amd import serve  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--base-port", type=int, default=5003)
    ap.add_argument("--nonce", default=os.getenv("MT_WEB_NONCE") or None, help="nonce the workers require (X-Nonce); generated when absent")
    ap.add_argument("--server", default=None, help="front server to register the workers with (its /register endpoint)")
    ap.add_argument("--server-nonce", default=None, help="the front server's nonce (defaults to --nonce)")
    ap.add_argument("--model-dir", default=None)
    ap.add_argument("--preload", action="store_true")
    a = ap.parse_args()
    wargs = (["--model-dir", a.model_dir] if a.model_dir else []) + (["--preload"] if a.preload else [])
    pool = serve.WorkerPool(host=a.host, base_port=a.base_port, nonce=a.nonce, worker_args=wargs)
    signal.signal(signal.SIGTERM, lambda *_: (pool.stop(), sys.exit(0)))
    try:
        pool.start()
        for inst in pool.executors.list:
            print(f"worker gpu={pool.gpus[inst.gpu]} {inst.url} nonce={pool.nonce}", flush=True)
        if a.server:
            pool.register_with(a.server, a.server_nonce or pool.nonce)
            print(f"registered {len(pool.executors.list)} workers with {a.server}", flush=True)
        while all(p.poll() is None for p in pool.procs):
            time.sleep(1.0)
        return 1
    except KeyboardInterrupt:
        return 0
    finally:
        pool.stop()


if __name__ == "__main__":
    sys.exit(main())
