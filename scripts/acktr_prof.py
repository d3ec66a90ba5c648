"""Development tool: steady-state timing of the batched ACKTR trainer (prints as it goes)."""
import sys, time
sys.path[:0] = ["deep-q-learning_tron_amd", "."]
import config  # noqa: F401  (sets MIOPEN_FIND_MODE before torch touches MIOpen)
import torch, ACKTR

for n, w, it in ((4096, 10, 4), (2048, 32, 2)):
    for rep in range(2):
        torch.cuda.reset_peak_memory_stats()
        t = time.perf_counter()
        out = ACKTR.train(n_envs=n, width=w, model="mul", reward="3", iterations=it, acktr=True)
        print(n, w, "rep", rep, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()
                                 if k not in ("brain", "last_stats")},
              "wall %.1fs peak %.1f GB" % (time.perf_counter() - t, torch.cuda.max_memory_allocated() / 1e9), flush=True)
