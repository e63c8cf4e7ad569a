"""The half-width chains on odd / tiny / ragged sizes against the exact fp32 chains (a robustness screen: no size is special-cased)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
T = torch.from_numpy
dev = torch.device("cuda:0")
def net(train):
    u = s2l.SimpleUnetLight().to(dev)
    u.load_state_dict({k[len("post_fusion_unet."):]: T(v) for k, v in W.make_unet_state_dict(0).items()})
    return u.train() if train else u.eval()
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))


def screen(log=print):
  bad = 0
  for (F, fh, fw) in [(1, 4, 4), (2, 5, 7), (3, 37, 53), (1, 33, 17), (2, 131, 77), (1, 63, 65), (5, 16, 500), (1, 501, 499),
                     (1, 530, 644)]:      # (the last: more pooling-adjoint blocks than the partial buffer holds: the separate reduce pass)
      x = T(np.random.default_rng(fh).random((F, fh, fw, 3), dtype=np.float32)).to(dev)
      d = T(np.random.default_rng(fw).standard_normal((F, fh, fw, 3)).astype(np.float32)).to(dev)
      # train mode, frozen
      r = {}
      for prec in ("fp32", "bf16h"):
          u = net(True)
          o, c = u.forward_train_frames_nhwc(x, precision=prec)
          r[prec] = (o, u.backward_train_frames(c, d))
      # train mode, training (parameter gradients)
      ut = net(True)
      o_t, c_t = ut.forward_for_backward(x, precision="bf16")
      g = {}
      dx_t = ut.backward_to_input(c_t, d, param_grads=g)
      fin = all(bool(torch.isfinite(v).all()) for v in g.values()) and bool(torch.isfinite(dx_t).all())
      same = bool(torch.equal(o_t, r["bf16h"][0]))
      # eval mode
      ue = net(False)
      oe32, ce32 = ue.forward_saved_nhwc(x)
      oeh, ceh = ue.forward_saved_nhwc(x, precision="bf16h")
      ge32, geh = ue.backward_input(ce32, d), ue.backward_input(ceh, d)
      ok = rel(r["bf16h"][0], r["fp32"][0]) < 0.08 and rel(oeh, oe32) < 0.03 and fin and same and bool(torch.isfinite(geh).all()) and bool(torch.isfinite(r["bf16h"][1]).all())
      bad += not ok
      log(f"{F}x{fh}x{fw}: train out rel {rel(r['bf16h'][0], r['fp32'][0]):.3e} dx rel {rel(r['bf16h'][1], r['fp32'][1]):.3f} | eval out rel {rel(oeh, oe32):.3e} dx rel {rel(geh, ge32):.3f} | "
            f"training route finite {fin}, same forward bits {same}  {'ok' if ok else 'BAD'}")
  return bad


if __name__ == "__main__":
    sys.exit(1 if screen() else 0)
