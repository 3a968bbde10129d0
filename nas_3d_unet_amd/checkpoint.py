"""Checkpoint / genotype interoperability with the reference; host logic only.

The reference saves `torch.save({...})` dictionaries (train.py:85-93: epoch, history, model_param, optim, scheduler,
best_loss; search.py:166-176 adds geno_count, optim_shell / optim_kernel, shell_scheduler / kernel_scheduler) and reads
them back in check_resume (train.py:52-67; search.py:108-127); the genotype travels as a pickled `(str(gene), count)`
that is `eval`ed (search.py:189-194; train.py:36-38).  Module state-dicts need no conversion (same keys and shapes).
What does need one is the optimizer: the trainers keep Adam's moments in flat buffers with ONE step counter, torch's
Adam keeps per-parameter tensors -- `adam_state_dict` / `load_adam_state_dict` translate both ways, and the plateau
schedulers export / import torch's ReduceLROnPlateau field names.
"""
from __future__ import annotations

import pickle

import torch

from .genotype import Genotype


# ------------------------------------------------------------------------------------------------ Adam
def _entries(fp, twin=None, real=None):
    """[(offset in the flat buffers, stored shape, name | None)] in the order torch.optim.Adam(real.parameters()) numbers its parameters.
    A trainer of a net with odd channel counts keeps the moments of the net's zero-padded TWIN (unet.PaddedTwin): its flat buffers are
    walked in the REAL module's parameter order and every moment goes through twin.extract / twin.embed_tensor, so the checkpoint holds
    the reference's shapes (a torch / reference checkpoint of the same net loads, and the other way round)."""
    if twin is None:
        return [(o, p.shape, None) for p, o in zip(fp.params, fp.offsets)]
    off = {id(p): (o, p.shape) for p, o in zip(fp.params, fp.offsets)}
    tp = dict(twin.twin.named_parameters())
    out = []
    for n, _ in real.named_parameters():
        if n in tp and id(tp[n]) in off:
            o, shape = off[id(tp[n])]
            out.append((o, shape, n))
    if len(out) != len(fp.params):
        raise ValueError("padded twin: %d of the trainer's %d parameters have no counterpart in the module" % (len(fp.params) - len(out), len(fp.params)))
    return out


def adam_state_dict(fp, lr, betas=(0.9, 0.999), eps=1e-8, twin=None, real=None):
    """torch.optim.Adam(params).state_dict() equivalent of a train.FlatParams (params in `fp.params` order; with a padded twin: in
    `real.parameters()` order and the reference's shapes)."""
    step = int(fp.step.item())
    state = {}
    ents = _entries(fp, twin, real)
    for i, (o, shape, name) in enumerate(ents):
        n = int(torch.Size(shape).numel())
        if step > 0:
            m, v = fp.exp_avg[o:o + n].view(shape), fp.exp_avg_sq[o:o + n].view(shape)
            if name is not None:
                m, v = twin.extract(name, m), twin.extract(name, v)
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
             "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
             "params": list(range(len(ents)))}
    return {"state": state, "param_groups": [group]}


def load_adam_state_dict(fp, sd, twin=None, real=None):
    """inverse of adam_state_dict; returns the learning rate stored in the checkpoint.
    Limit: the flat Adam keeps ONE step counter, so every parameter with state must be at the same step (true for every
    checkpoint the reference writes: all parameters receive a gradient in every step); otherwise ValueError."""
    steps = set()
    fp.exp_avg.zero_()
    fp.exp_avg_sq.zero_()
    for i, (o, shape, name) in enumerate(_entries(fp, twin, real)):
        st = sd["state"].get(i)
        if st is None:
            continue
        n = int(torch.Size(shape).numel())
        m, v = st["exp_avg"].to(fp.exp_avg.device), st["exp_avg_sq"].to(fp.exp_avg.device)
        if name is not None:     # reference shapes -> the twin's (zeros at the padded entries, which Adam never moves: masked gradients)
            m, v = twin.embed_tensor(name, m, shape), twin.embed_tensor(name, v, shape)
        fp.exp_avg[o:o + n].copy_(m.reshape(-1))
        fp.exp_avg_sq[o:o + n].copy_(v.reshape(-1))
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError("checkpoint has different Adam step counts per parameter: %s" % sorted(steps))
    fp.step.fill_(steps.pop() if steps else 0)
    return float(sd["param_groups"][0]["lr"])


# ------------------------------------------------------------------------------------------------ scheduler
_SCHED_FIELDS = ("factor", "patience", "threshold", "cooldown", "eps", "best", "num_bad_epochs", "cooldown_counter", "last_epoch")


def scheduler_state_dict(s):
    """ReduceLROnPlateau.state_dict() field names for a train.PlateauLR"""
    d = {k: getattr(s, k) for k in _SCHED_FIELDS}
    d.update({"mode": "min", "threshold_mode": "rel", "min_lrs": [s.min_lr], "mode_worse": float("inf"), "_last_lr": [s.get_lr()]})
    return d


def load_scheduler_state_dict(s, d):
    for k in _SCHED_FIELDS:
        if k in d:
            setattr(s, k, d[k])
    if "min_lrs" in d:
        s.min_lr = d["min_lrs"][0]


# ------------------------------------------------------------------------------------------------ train.py:85-93 / 52-67
def _checked(trainer):
    """a checkpoint is a host-visible point: a timed-out stream hand-off (whose update was withheld on the device) raises here
    instead of being written out as if the steps since had trained (train.Trainer.check_sync)"""
    chk = getattr(trainer, "check_sync", None)
    if chk is not None:
        chk()


def _twin_of(trainer, real):
    """(twin, real module) of a trainer that trains a padded twin, (None, None) otherwise"""
    tw = getattr(trainer, "_twin", None)
    return (tw, real) if tw is not None else (None, None)


def train_state_dicts(trainer, epoch, history, best_loss):
    _checked(trainer)
    return {"epoch": epoch, "history": history, "model_param": trainer.model.state_dict(),
            "optim": adam_state_dict(trainer.fp, trainer.lr, trainer.betas, trainer.eps, *_twin_of(trainer, trainer.model)),
            "scheduler": scheduler_state_dict(trainer.scheduler), "best_loss": best_loss}


def load_train_state_dicts(trainer, sd, new_lr=False):
    """returns (next epoch, history, best_loss) like check_resume"""
    trainer.model.load_state_dict(sd["model_param"])   # parameters are views of the flat buffer: copied in place
    getattr(trainer, "sync_from_module", lambda: None)()   # (a padded twin re-embeds the loaded parameters)
    if not new_lr:
        trainer.set_lr(load_adam_state_dict(trainer.fp, sd["optim"], *_twin_of(trainer, trainer.model)))
        load_scheduler_state_dict(trainer.scheduler, sd["scheduler"])
    return sd["epoch"] + 1, sd["history"], sd["best_loss"]


# ------------------------------------------------------------------------------------------------ search.py:166-176 / 108-127
def search_state_dicts(trainer, epoch, geno_count, history, best_loss):
    """the reference's search checkpoint: two optimizers, two schedulers (the reference stores the KERNEL scheduler under
    both scheduler keys, search.py:174; here each key holds its own scheduler)"""
    _checked(trainer)
    return {"epoch": epoch, "geno_count": geno_count, "history": history, "model_param": trainer.model.state_dict(),
            "optim_shell": adam_state_dict(trainer.afp, trainer.lr_shell, trainer.betas, trainer.eps),
            "optim_kernel": adam_state_dict(trainer.fp, trainer.lr_kernel, trainer.betas, trainer.eps, *_twin_of(trainer, trainer.model.kernel)),
            "kernel_scheduler": scheduler_state_dict(trainer.kernel_scheduler),
            "shell_scheduler": scheduler_state_dict(trainer.shell_scheduler), "best_loss": best_loss}


def load_search_state_dicts(trainer, sd, new_lr=False):
    """returns (next epoch, geno_count, history, best_loss) like search.py's check_resume"""
    trainer.model.load_state_dict(sd["model_param"])   # kernel weights and alphas are views of the flat buffers: copied in place
    getattr(trainer, "sync_from_module", lambda: None)()   # (a padded twin re-embeds the loaded parameters)
    if not new_lr:
        trainer.set_shell_lr(load_adam_state_dict(trainer.afp, sd["optim_shell"]))
        trainer.set_kernel_lr(load_adam_state_dict(trainer.fp, sd["optim_kernel"], *_twin_of(trainer, trainer.model.kernel)))
        load_scheduler_state_dict(trainer.shell_scheduler, sd["shell_scheduler"])
        load_scheduler_state_dict(trainer.kernel_scheduler, sd["kernel_scheduler"])
    return sd["epoch"] + 1, sd["geno_count"], sd["history"], sd["best_loss"]


# ------------------------------------------------------------------------------------------------ genotype pickle
def save_genotype(path, gene, count=1):
    """search.py:189-194: pickle of (str(gene), count)"""
    with open(path, "wb") as f:
        pickle.dump((str(gene), count), f)


class _StrTupleUnpickler(pickle.Unpickler):
    """a genotype file holds (str, int): no class needs to be importable, so none is"""

    def find_class(self, module, name):
        raise pickle.UnpicklingError("genotype file references %s.%s: only a (str, count) tuple is accepted" % (module, name))


def load_genotype(path):
    """train.py:36-38 does `eval(pickle.load(f)[0])`; here the pickle may only contain builtin containers / str / int and
    the text is parsed as a literal `Genotype(down=[...], up=[...])` call (ast), never evaluated"""
    import ast
    with open(path, "rb") as f:
        text = _StrTupleUnpickler(f).load()[0]
    try:
        call = ast.parse(text.strip(), mode="eval").body
        if not (isinstance(call, ast.Call) and isinstance(call.func, ast.Name) and call.func.id == "Genotype"):
            raise ValueError
        args = [ast.literal_eval(a) for a in call.args]
        kw = {k.arg: ast.literal_eval(k.value) for k in call.keywords}
        gene = Genotype(*args, **kw)
    except (ValueError, SyntaxError, TypeError):
        raise ValueError("not a Genotype: %r" % (text,))
    return gene
