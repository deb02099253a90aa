"""Do S independent learners (own envs, replay rings, networks, Philox keys, hipGraph) overlap on ONE MI355X when every
learner replays its graph on its own HIP stream?  Each launch of the lock-step iteration occupies <= 64 of the 256 CUs
and mostly waits, so the reference's unit of parallelism -- the seed loop (scripts/navigation1.sh:4-8) -- is packed on
the device.  Prints aggregate env-steps/s and grad-steps/s for S in {1, 2, 4, 8}.
    python profiles/seed_pack_probe.py [steps=400] [updates_per_step=1]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recovery_rl_amd import runtime  # noqa: E402

runtime.configure(graph_packet_capture=int(os.environ.get("PACK_PACKET_CAPTURE", "0")), log=False)
import torch  # noqa: E402

import arg_utils  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
U = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
out = []
for S in [int(x) for x in os.environ.get("PACK_S", "1,2,4,8").split(",")]:
    loops, streams = [], []
    for k in range(S):
        cfg = arg_utils.get_args(bench.config_argv("navigation1", 1 + k, 4096, U))
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            loop = bench.build_loop(cfg, dev)
            loop.step_outputs = False
            loop.capture(online_qrisk=True)
        loops.append(loop)
        streams.append(st)
    torch.cuda.synchronize()

    def run(n):
        if os.environ.get("PACK_THREADS", "0") == "1":          # one host thread per learner
            import threading

            def work(loop, st):
                with torch.cuda.stream(st):
                    for _ in range(n):
                        loop.replay()
            ts = [threading.Thread(target=work, args=(l, s)) for l, s in zip(loops, streams)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            return
        for _ in range(n):
            for loop, st in zip(loops, streams):
                with torch.cuda.stream(st):
                    loop.replay()
    run(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    upd = [int(l.agent.fast.critic.step[0].item()) for l in loops]
    rec = {"packet_capture": os.environ.get("PACK_PACKET_CAPTURE", "0"), "threads": os.environ.get("PACK_THREADS", "0"),
           "seeds_per_gpu": S, "updates_per_step": U, "ms_per_round": dt / steps * 1e3,
           "aggregate_env_steps_per_s": S * 4096 * steps / dt, "aggregate_sac_grad_steps_per_s": S * U * steps / dt,
           "device_update_counters": upd}
    out.append(rec)
    print(rec, file=sys.stderr)
    del loops, streams
    torch.cuda.empty_cache()
base = out[0]["aggregate_env_steps_per_s"]
for r in out:
    r["speedup_vs_one_seed"] = r["aggregate_env_steps_per_s"] / base
print(json.dumps(out))
