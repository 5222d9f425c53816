import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from rtx_nerf_amd import train
torch.cuda.set_device(0)
orig_flush = train.Trainer.flush_captured
def flush(self):
    torch.cuda.synchronize()
    print("before flush: loss", float(self.loss.item()), "pending", self._g_pending, "step", self.step_count, "total_host",
          [int(s["total_host"][0]) for s in self._g_sets], "cap", self._g_cap)
    l = orig_flush(self)
    torch.cuda.synchronize()
    print("after flush: loss", float(self.loss.item()), "pixels finite", bool(torch.isfinite(self.pixels).all()),
          "params finite", bool(torch.isfinite(self.master).all()), "radiance finite", bool(torch.isfinite(self.radiance[:self._g_cap * 32]).all()),
          "targets", float(self._g_sets[0]["targets"].abs().max()), float(self._g_sets[1]["targets"].abs().max()))
    k = 1
    for name, fn in (("flush graph again", lambda: self._graphs["flush"][k].replay()), ("eager _captured_gradients", lambda: self._captured_gradients(k)),
                     ("flush graph set 0", lambda: self._graphs["flush"][0].replay()), ("eager set 0", lambda: self._captured_gradients(0))):
        fn()
        torch.cuda.synchronize()
        print(name, "-> loss", float(self.loss.item()), "total", [int(s_["total"].item()) for s_ in self._g_sets],
              "num_stored sum", [int(s_["num_stored"][:self._g_n].sum().item()) for s_ in self._g_sets],
              "indices max", [int(s_["indices"][:self._g_n].max().item()) for s_ in self._g_sets])
    return l
train.Trainer.flush_captured = flush
if len(sys.argv) > 1 and sys.argv[1] == "nostages":
    train.Trainer.time_stages = lambda self, *a, **k: {}
r = bench.train_ref_record(4096, 128, 60, 5, dense_grid=False, mode="nerf")
print({k: r[k] for k in ("ms_per_step", "ms_per_step_host_count", "loss_last", "loss_last_captured", "truncated_steps", "segment_capacity")})
