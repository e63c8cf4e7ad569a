"""`torch.optim.Adam` for the drop-in trainer as ONE kernel launch per parameter group.

The reference's loop builds `optim.Adam(model.parameters(), lr=...)` (train.py:166) and steps it once per iteration after `check_weights`
(training.py:559-574).  torch's foreach path is eleven multi-tensor launches over the 42 + 32 tensors of the May model plus their host-side
list handling (~0.5 ms of a 1.4-ms-per-frame step); `FusedAdam.step()` is one launch of `s2l_adam_step` (csrc/train.hip): torch's
`_single_tensor_adam` arithmetic operation for operation in fp32, every tensor of the group through a device table of pointers, and -- because
the pass reads every parameter anyway -- the NaN scan of `check_weights` (src/common.py:56-64) folded into it (`nan_flags`).

Same constructor arguments, same `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter) as `torch.optim.Adam`: a checkpoint
written by either loads into the other (the reference's checkpoints carry `optimizer.state_dict()`, checkpoints.py:38-41).  fp32 CUDA
parameters with dense gradients; no amsgrad / maximize / capturable (the reference uses none of them).  There is no CPU path."""
from __future__ import annotations

import ctypes

import torch

from . import _abi


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("FusedAdam: amsgrad is not part of the reference's optimizer (train.py:166)")
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._plans = {}           # per group index: the cached launch plan (block table, counts, pinned staging, device buffers)
        self.nan_flags = None      # [(parameter, bool)] of the last step(): which parameters held a NaN BEFORE the update (after `nan_report()`)
        self._flag_jobs = []

    # ---- launch plan of one group: everything that only depends on the tensors' sizes
    def _plan(self, gi, ps):
        key = tuple((p.numel(), p.device) for p in ps)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        lib = _abi.load()
        dev = ps[0].device
        chunk = int(lib.s2l_adam_chunk())
        blocks = []
        for t, p in enumerate(ps):
            blocks += [(t, off) for off in range(0, p.numel(), chunk)]
        plan = {"key": key, "n_blocks": len(blocks),
                "blocks": torch.tensor(blocks, dtype=torch.int32).reshape(-1, 2).to(dev),
                "counts": torch.tensor([p.numel() for p in ps], dtype=torch.int64).to(dev),
                "host": torch.empty(len(ps) * 4, dtype=torch.int64, pin_memory=True),          # {param, grad, exp_avg, exp_avg_sq} pointers
                "table": torch.empty(len(ps) * 4, dtype=torch.int64, device=dev),
                "ptrs": None, "copied": None}
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _abi.load()
        self._flag_jobs = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            for p in ps:
                if p.device != dev or dev.type != "cuda" or p.dtype != torch.float32 or p.grad.is_sparse or not p.is_contiguous():
                    raise _abi.S2LError("FusedAdam takes contiguous fp32 CUDA parameters of one device with dense gradients (no CPU path)")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)          # (torch.optim.Adam's own layout: a host scalar tensor)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps = {float(self.state[p]["step"]) for p in ps}
            if len(steps) != 1:
                raise _abi.S2LError("FusedAdam: the parameters of one group must share their step count (one launch has one bias correction)")
            step_no = int(steps.pop()) + 1
            plan = self._plan(gi, ps)
            grads = [p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.to(torch.float32).contiguous() for p in ps]
            ptrs = []
            for p, g in zip(ps, grads):
                st = self.state[p]
                ptrs += [p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()]
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream()
                if ptrs != plan["ptrs"]:      # (gradients are fresh tensors every step in the fused trainer: 2.4 KB of pointers cross once per step)
                    if plan["copied"] is not None:
                        plan["copied"].synchronize()      # the pinned staging block of the previous step has left
                    plan["host"].copy_(torch.tensor(ptrs, dtype=torch.int64))
                    plan["table"].copy_(plan["host"], non_blocking=True)
                    plan["copied"] = torch.cuda.Event()
                    plan["copied"].record(stream)
                    plan["ptrs"] = ptrs
                # (flags per step, not per plan: a pipelined caller reads step k's report after step k + 1 has been queued; the pinned block
                #  comes from torch's caching host allocator)
                flags = torch.zeros(len(ps), dtype=torch.int32, device=dev)
                flags_host = torch.empty(len(ps), dtype=torch.int32, pin_memory=True)
                b1, b2 = group["betas"]
                _abi.check(lib.s2l_adam_step(ctypes.c_void_p(plan["table"].data_ptr()), ctypes.c_void_p(plan["blocks"].data_ptr()),
                                             ctypes.c_void_p(plan["counts"].data_ptr()), len(ps), plan["n_blocks"], float(group["lr"]), float(b1),
                                             float(b2), float(group["eps"]), float(group["weight_decay"]), step_no,
                                             ctypes.c_void_p(flags.data_ptr()), ctypes.c_void_p(stream.cuda_stream)), "s2l_adam_step")
                flags_host.copy_(flags, non_blocking=True)
                done = torch.cuda.Event()
                done.record(stream)
            self._flag_jobs.append((ps, flags_host, done, grads))      # (`grads` keeps converted copies alive until the launch has run)
            for p in ps:
                self.state[p]["step"] += 1
                # the kernel wrote through raw pointers: tell autograd the tensor changed in place, as torch's own in-place update would --
                # the model's packed-weight caches (TalkingFace.packed_weights, SimpleUnetLight.packed_weights*) key on the version counters
                torch.autograd.graph.increment_version(p)
        return loss

    def take_nan_jobs(self):
        """The last `step()`'s flag transfers, handed over to the caller (Trainer keeps them with its pending step): pass to `nan_report`."""
        jobs, self._flag_jobs = self._flag_jobs, []
        return jobs

    def nan_report(self, jobs=None):
        """[(parameter, True / False)] for the parameters of the last `step()` (or of `jobs`): held a NaN before that update.  Waits for that
        step's flags (a few bytes behind the kernel) -- call it after the next piece of work has been queued."""
        out = []
        for ps, host, done, _ in (self._flag_jobs if jobs is None else jobs):
            done.synchronize()
            out += list(zip(ps, [bool(v) for v in host.tolist()]))
        if jobs is None:
            self._flag_jobs = [(ps, host, done, None) for ps, host, done, _ in self._flag_jobs]
        self.nan_flags = out
        return out

    def covers(self):
        """The parameters whose NaN state `nan_report()` answers for (ids), given the gradients present now."""
        return {id(p) for g in self.param_groups for p in g["params"] if p.grad is not None}
