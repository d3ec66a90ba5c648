"""Development tool: timing of the batched ACKTR trainer at BASELINE config 5's size (prints as it goes)."""
import sys, time
sys.path[:0] = ["deep-q-learning_tron_amd", "."]
import config  # noqa: F401  (sets MIOPEN_FIND_MODE before torch touches MIOpen)
import torch, ACKTR

n, w = int(sys.argv[1]), int(sys.argv[2])
it = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mb = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
torch.cuda.reset_peak_memory_stats()
t = time.perf_counter()
out = ACKTR.train(n_envs=n, width=w, model="mul", reward="3", iterations=it, acktr=True, log_every=1,
                  micro_batch=mb, act_batch=min(mb, 16384))
print(n, w, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items() if k not in ("brain", "last_stats")},
      "wall %.1fs peak %.1f GB" % (time.perf_counter() - t, torch.cuda.max_memory_allocated() / 1e9), flush=True)
