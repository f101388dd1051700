"""Checkpoint save / resume in the reference's on-disk layout (train_gen.py:252-290, 345-357).

A checkpoint is `torch.save({'model_state_dict', 'scheduler_state_dict', 'optimizer_state_dict', 'iter_id'})`; the model
part is `EncoderDecoderModel.state_dict()` with the reference's 861 keys, so files written by the reference load
here and files written here load there.  The optimizer part is either the flat AdamW moments of `FusedAdamW` (compact:
two tensors) or, with `reference_format=True`, the reference's own per-tensor `optimizer.state_dict()` layout (one param
group per tensor in the order train_gen.py:209-245 builds them, incl. the slots of the decoder embeddings orphaned by the
aliasing: optim.reference_param_index) -- so `-continue` (train_gen.py:254-276) works in both directions.  Loading detects
the layout.
"""
import torch


def _cpu(o):
    if torch.is_tensor(o):
        return o.detach().cpu()
    if isinstance(o, dict):
        return {k: _cpu(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_cpu(v) for v in o)
    return o


def save_checkpoint(path, model, optimizer=None, iter_id=0, reference_format=False):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ck = {"model_state_dict": sd, "iter_id": int(iter_id), "scheduler_state_dict": {}, "optimizer_state_dict": {}}
    if optimizer is not None:
        st = optimizer.export_reference_state() if reference_format else optimizer.state_dict()
        ck["optimizer_state_dict"] = _cpu(st)
        # the keys utils/optim_utils.py's _LRScheduler subclass saves that matter for a resume
        ck["scheduler_state_dict"] = {"last_epoch": optimizer.sched_step, "_step_count": optimizer.sched_step + 1,
                                      "warmup_steps": optimizer.warmup_steps, "t_total": optimizer.t_total,
                                      "min_lr": optimizer.min_lr}
        if reference_format:
            ck["scheduler_state_dict"]["base_lrs"] = [g["initial_lr"] for g in st["param_groups"]]
    torch.save(ck, path)


def load_checkpoint(path, model, optimizer=None, cont=True, map_location="cpu"):
    """cont=True : train_gen.py:254-276 (`-continue`): key-intersected load of the whole model (+ optimizer when given);
    cont=False: train_gen.py:278-289: only the ENCODER's keys are taken (how checkpoints/basemodel is ingested).
    Returns iter_id (0 when absent)."""
    ck = torch.load(path, map_location=map_location)
    sd = ck["model_state_dict"] if "model_state_dict" in ck else ck
    if cont:
        own = model.state_dict()
        own.update({k: v for k, v in sd.items() if k in own})
        model.load_state_dict(own)
        if optimizer is not None and ck.get("optimizer_state_dict"):
            st = ck["optimizer_state_dict"]
            sch = ck.get("scheduler_state_dict") or {}
            if "last_epoch" in sch:
                optimizer.sched_step = int(sch["last_epoch"])
            if "m" in st:
                st = dict(st, sched_step=optimizer.sched_step if "last_epoch" in sch else st.get("sched_step", 0))
                optimizer.load_state_dict(st)
            elif "state" in st and "param_groups" in st:          # written by the reference (or reference_format=True)
                optimizer.import_reference_state(st)
            elif "reference_state" in st:                         # a reference-format state saved before it was ever applied
                optimizer.load_state_dict(st)
        return int(ck.get("iter_id", 0)) if isinstance(ck, dict) else 0
    enc = model.encoder
    own = enc.state_dict()
    # reference keys are relative to dialog_encoder (no 'encoder.' prefix) in VisDial-BERT base models, or carry it
    take = {}
    for k, v in sd.items():
        kk = k[len("encoder."):] if k.startswith("encoder.") else k
        if kk in own:
            take[kk] = v
    if not take:
        raise RuntimeError("no encoder keys found in %s" % path)
    own.update(take)
    enc.load_state_dict(own)
    return 0
