"""BaseModel: restore()/save() surface of the reference (models/base_model.py:8-26 on top of dg_util's BaseModel).

dg_util is not vendored by the reference; the semantics used here are the ones its call sites force:
``restore`` loads the newest checkpoint of ``args.checkpoint_dir`` and returns the iteration encoded in its file name,
``save`` writes ``<iteration>.pt`` and keeps the newest ``num_to_keep`` files (-1 = all).
"""
import glob
import os
import re

import torch
from torch import nn

from .. import constants


def _ckpt_iteration(path):
    m = re.search(r"(\d+)\.pt$", os.path.basename(path))
    return int(m.group(1)) if m else -1


def save_checkpoint(model, directory, num_to_keep, iteration):
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "%09d.pt" % iteration)
    tmp = path + ".tmp%d" % os.getpid()
    torch.save({k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}, tmp)
    os.replace(tmp, path)   # readers never see a half-written checkpoint
    if num_to_keep is not None and num_to_keep > 0:
        files = sorted(glob.glob(os.path.join(directory, "*.pt")), key=_ckpt_iteration)
        for old in files[:-num_to_keep]:
            try:
                os.remove(old)
            except FileNotFoundError:   # another process pruned it first
                pass
    return path


class BaseModel(nn.Module):
    def __init__(self, args):
        super(BaseModel, self).__init__()
        self.args = args
        self.saves = 0

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def restore(self, skip_filter=None) -> int:
        iteration = 0
        if getattr(self.args, "restore", False):
            directory = self.args.checkpoint_dir
            files = []
            for root, _, names in os.walk(directory) if os.path.isdir(directory) else []:
                files += [os.path.join(root, n) for n in names if n.endswith(".pt")]
            if not files:
                print("No checkpoint found in", directory)
                return 0
            newest = max(files, key=_ckpt_iteration)
            state = torch.load(newest, map_location="cpu")
            saved_prefixes = [p for p in (getattr(self.args, "saved_variable_prefix", None) or []) if p]
            new_prefixes = [p for p in (getattr(self.args, "new_variable_prefix", None) or []) if p]
            renamed = {}
            for key, val in state.items():
                for sp, np_ in zip(saved_prefixes, new_prefixes):
                    if key.startswith(sp):
                        key = np_ + key[len(sp):]
                        break
                if skip_filter is not None and skip_filter(key):
                    continue
                renamed[key] = val
            missing, unexpected = self.load_state_dict(renamed, strict=False)
            print("Restored", newest, "missing", len(missing), "unexpected", len(unexpected))
            iteration = max(_ckpt_iteration(newest), 0)
        return iteration

    def save(self, iteration, num_to_keep=1):
        if getattr(self.args, "save", False):
            save_checkpoint(self, os.path.join(self.args.checkpoint_dir, constants.TIME_STR), num_to_keep, iteration)
            long_freq = getattr(self.args, "long_save_frequency", 0)
            if long_freq and self.saves > 0 and self.saves % long_freq == 0:
                save_checkpoint(self, self.args.long_save_checkpoint_dir, -1, iteration)
            self.saves += 1
