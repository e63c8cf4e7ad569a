"""A cached view of an `nn.Module` tree for the per-step host code.

`dict(model.named_parameters())` and `model.train()` walk the whole tree -- ~90 modules with the post-fusion U-Net -- through
generators: 0.25 - 0.4 ms per call, and a one-frame training step made four of them (about a fifth of its host time).  Here the tree
is walked once; every later call checks, module by module, that the children and the number of registered parameters and buffers are still the
ones seen (identity comparisons, ~25 us) and rebuilds otherwise.  The parameter OBJECTS are looked up in the owning modules' own
`_parameters` dictionaries at every call, so a parameter that was re-assigned (`m.fc.weight = nn.Parameter(...)`) is found."""
import torch.nn as nn


def _build(root: nn.Module):
    mods = list(root.named_modules())
    tree = {
        "kids": [(m, tuple(m._modules.values()), len(m._parameters) + len(m._buffers)) for _, m in mods],
        "params": [((f"{prefix}." if prefix else "") + attr, m._parameters, attr) for prefix, m in mods for attr in m._parameters],
        "buffers": [((f"{prefix}." if prefix else "") + attr, m._buffers, attr, m._non_persistent_buffers_set) for prefix, m in mods
                    for attr in m._buffers],
        "plain_state": all(type(m)._save_to_state_dict is nn.Module._save_to_state_dict and not m._state_dict_hooks for _, m in mods),
        "plain_train": all(type(m).train is nn.Module.train for _, m in mods),
    }
    root.__dict__["_s2l_tree"] = tree
    return tree


def _tree(root: nn.Module):
    tree = root.__dict__.get("_s2l_tree")
    if tree is not None:
        for m, kids, n_par in tree["kids"]:
            if len(m._parameters) + len(m._buffers) != n_par or tuple(m._modules.values()) != kids:
                tree = None
                break
    return tree if tree is not None else _build(root)


def param_map(root: nn.Module) -> dict:
    """{name: parameter}: what `dict(root.named_parameters())` returns (shared parameters appear once, under their first name)."""
    out, seen = {}, set()
    for name, owner, attr in _tree(root)["params"]:
        p = owner[attr]
        if p is not None and id(p) not in seen:
            seen.add(id(p))
            out[name] = p
    return out


def set_training(root: nn.Module, mode: bool = True) -> nn.Module:
    """`root.train(mode)`; the flags are set directly when no module of the tree overrides `train`."""
    tree = _tree(root)
    if not tree["plain_train"]:
        return root.train(mode)
    for m, _, _ in tree["kids"]:
        m.training = mode
    return root


def state_tensors(root: nn.Module):
    """[(key, tensor)] of `root.state_dict()`: parameters, then persistent buffers (the order within is the tree's, not state_dict's
    interleaving, and a module reachable under two names contributes its tensors once -- callers that need state_dict()'s exact keys
    use state_dict()).  Falls back to state_dict() when a module customises it."""
    tree = _tree(root)
    if not tree["plain_state"]:
        return list(root.state_dict().items())
    out = [(name, owner[attr]) for name, owner, attr in tree["params"] if owner[attr] is not None]
    out += [(name, owner[attr]) for name, owner, attr, skip in tree["buffers"] if owner[attr] is not None and attr not in skip]
    return out


class TensorSlots:
    """The parameters / buffers `names` (dotted paths under `root`) resolved to (owning dictionary, key) pairs once, plus every
    parent -> child link on the way there.  `valid()` re-checks those links by identity (a few dozen dictionary look-ups, ~3 us), so a
    sub-module replaced at ANY depth -- `unet.inc.double_conv[1] = ...`, `nn.SyncBatchNorm.convert_sync_batchnorm(unet)`,
    `model.audio_net.encoder_conv[0] = ...`, `add_module`, `del` -- is noticed and the slots are resolved again; a tensor re-assigned
    inside an unchanged module is found because the dictionary is the module's own."""
    __slots__ = ("links", "slots")

    def __init__(self, root: nn.Module, names):
        self.links, self.slots = [], []
        seen = set()
        for full in names:
            path, _, attr = full.rpartition(".")
            m = root
            for key in (path.split(".") if path else ()):
                child = m._modules[key]
                if (id(m), key) not in seen:
                    seen.add((id(m), key))
                    self.links.append((m._modules, key, child))
                m = child
            if attr in m._parameters:
                self.slots.append((m._parameters, attr))
            elif attr in m._buffers:
                self.slots.append((m._buffers, attr))
            else:
                raise KeyError(f"{type(root).__name__} has no parameter or buffer {full!r}")

    def valid(self) -> bool:
        for d, k, c in self.links:
            if d.get(k) is not c:
                return False
        return True

    def tensors(self):
        return [d[a] for d, a in self.slots]


def tensor_slots(root: nn.Module, names, key: str = "_s2l_slots"):
    """`[tensor for name in names]` through a `TensorSlots` kept in `root.__dict__[key]` and rebuilt when the tree changed."""
    slots = root.__dict__.get(key)
    if slots is None or not slots.valid():
        slots = root.__dict__[key] = TensorSlots(root, names)
    try:
        return slots.tensors()
    except KeyError:      # a parameter was deleted and re-registered as the other kind (parameter <-> buffer): resolve again
        slots = root.__dict__[key] = TensorSlots(root, names)
        return slots.tensors()
