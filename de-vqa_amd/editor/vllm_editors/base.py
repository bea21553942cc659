"""Editor plugin ABCs: the API of R/editor/vllm_editors/base.py.

`VLLMBaseEditor` (base.py:20-63) is the plugin interface every editor implements; `VLLMBaseEditorWithTraining`
(base.py:67-268) adds the training surface of hyper-network editors -- the abstract hooks, `train_init(vllm_edit_data,
batch_size, ...)`, `train(total_epochs)`, `save_ckpt(i, epoch, loss, ema_loss)`, `load_ckpt(ckpt_path, restrict, load_opt)`,
`set_random_seeds`, `write_logs` -- with the reference's names, argument order and defaults, the `Best` checkpoint layout,
the records/<editor>/<model>/<train name>/{checkpoints,logs,config.yaml} directory layout and the EMA-best save rule, so a
trainable editor written against the reference ABC drops in (and `train_vllm_editor.py` calls `train_init` then `train`
exactly as R/train_vllm_editor.py:85-89).

Differences in HOW, not WHAT:
  * "train modules" and the optimizer only need `state_dict()` / `load_state_dict()`: `nn.Module`s and torch optimizers
    satisfy that, and so do the plain device-tensor holders of editors whose arithmetic is in HIP (MEND_VL);
  * checkpoints are read with `torch.load(..., weights_only=True)` (nothing in the file is executed);
  * TensorBoard is optional: without the `tensorboard` package the scalars go to logs/scalars.jsonl through a writer with the
    same `add_scalar(name, value, step)` call;
  * `ParallelDataset` may prefetch on a second HIP stream of the same GPU (`data_prefetch_device`) instead of a second GPU.
"""
import json
import os
from abc import ABC, abstractmethod
from copy import deepcopy
from dataclasses import asdict, is_dataclass
from datetime import datetime
from typing import Dict, List, Tuple, Union

import numpy as np
import torch
import yaml

from ..base import BaseConfig
from ..vllms_for_edit.base import BaseVLLMForEdit


class VLLMBaseEditor(ABC):
    def __init__(self, vllm: BaseVLLMForEdit, device="cuda"):
        if not isinstance(vllm, BaseVLLMForEdit):
            raise RuntimeError("vllm must be a BaseVLLMForEdit")
        self.vllm = vllm
        self.vllm.set_device(device)
        self.device = device if device != "auto" else "cuda:0"
        assert self.if_model_decoder_only()  # only decoder-only llms are supported (base.py:26)

    def if_model_decoder_only(self) -> bool:
        return not self.vllm.model.config.is_encoder_decoder

    @abstractmethod
    def name_of_editor_and_model(self) -> Tuple[str, str]:
        """-> (editor_name, model_name)"""

    @abstractmethod
    def restore_to_original_model(self):
        """restore the original weights after editing"""

    @abstractmethod
    def edit_one_piece(self, request: Dict):
        """request = {'image': path|None, 'prompt': str, 'target_new': str, ...}"""

    @abstractmethod
    def edit_batch(self, requests: List[Dict]):
        """list of requests"""

    @abstractmethod
    def if_can_batch_edit(self) -> bool:
        pass


class JsonlScalarWriter:
    """`SummaryWriter.add_scalar(tag, value, step)` without TensorBoard: one JSON object per line in <logs>/scalars.jsonl."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self._f = open(self.path, "a")

    def add_scalar(self, tag, value, step):
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
        self._f.flush()

    def close(self):
        self._f.close()


def _summary_writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter   # needs the `tensorboard` package
        return SummaryWriter(log_dir)
    except Exception:
        return JsonlScalarWriter(log_dir)


class VLLMBaseEditorWithTraining(VLLMBaseEditor):
    def __init__(self, vllm: BaseVLLMForEdit, config: BaseConfig, device="cuda"):
        super().__init__(vllm, device)
        self.cfg = config
        self.log_writer = None

    # ---- hooks an editor implements (base.py:72-119) ---------------------------------------------------------------
    @abstractmethod
    def get_modules_for_training(self) -> Dict[str, object]:
        """{name: object with state_dict() / load_state_dict(sd, strict)} -- what save_ckpt / load_ckpt store."""

    @abstractmethod
    def reinit_train_parameters(self):
        """Reinitialize parameters of modules to be trained."""

    @abstractmethod
    def preprocess_train_data(self, vllm_edit_data) -> List:
        """Raw training data -> the list batches are sampled from every iteration."""

    @abstractmethod
    def organize_batch_data(self, a_batch_of_training_data: List):
        """A batch (list) of training data -> what train_a_batch consumes; called by the data generator."""

    @abstractmethod
    def train_a_batch(self, a_batch_of_organized_training_data):
        """Train the editor once.  -> (loss: float, log_dict: Dict)"""

    @abstractmethod
    def get_a_new_optimizer(self):
        """-> opt, or (opt, lr_scheduler); both with state_dict() / load_state_dict()."""

    @abstractmethod
    def set_train(self, is_train: bool):
        """Set training state for editor."""

    def data_prefetch_device(self):
        """Device whose second HIP stream the data generator's producer thread may use (None: no stream switch)."""
        return None

    # ---- base.py:121-140 ----------------------------------------------------------------------------------------------
    def set_random_seeds(self, seed: int):
        import random
        import time
        if seed is None:
            seed = int(time.time() * 10000) % 99999999
        print("Random seed is", seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed)
            torch.cuda.manual_seed_all(seed)
        np.random.seed(seed)
        random.seed(seed)
        self.random_seed = seed

    def other_train_init_final(self):
        """Called at the end of `self.train_init`."""

    def other_train_init_begin(self):
        """Called at the begin of `self.train_init`."""

    # ---- base.py:142-191 ----------------------------------------------------------------------------------------------
    def train_init(self, vllm_edit_data, batch_size: int, records_dir: str = "records", train_name_prefix: str = None,
                   train_name: str = None, load_ckpt_path: str = None, save_ckpt_per_i: int = 3000, log_per_i: int = 10,
                   ema_alpha: float = 0.1, random_seed: int = None, data_buffer_size=8,
                   seed_init_train_params_if_no_ckpt_path=True):
        """Initialises the data generator `self.data_generator`, the checkpoint / log directory, the writer and the optimizer."""
        from ...dataset import ParallelDataset
        from ...dataset.vllm import BaseVLLMEditData
        self.set_random_seeds(random_seed)
        self.other_train_init_begin()

        def get_data_by_ids_func(ids):
            return self.organize_batch_data([training_data[i] for i in ids])
        assert isinstance(vllm_edit_data, BaseVLLMEditData)
        training_data = self.preprocess_train_data(vllm_edit_data)
        self.data_generator = ParallelDataset(len(training_data), get_data_by_ids_func, batch_size, True, data_buffer_size, False,
                                              self.random_seed, True, device=self.data_prefetch_device())
        t = datetime.now().strftime("%Y.%m.%d-%H.%M.%S")
        train_name = (train_name_prefix + "-" if train_name_prefix else "") + (train_name if train_name else t)
        records_dir = os.path.join(records_dir, *self.name_of_editor_and_model(), train_name)
        self.save_ckpt_dir = os.path.join(records_dir, "checkpoints")
        os.makedirs(self.save_ckpt_dir, exist_ok=True)
        logs_path = os.path.join(records_dir, "logs")
        os.makedirs(logs_path, exist_ok=True)
        with open(os.path.join(records_dir, "config.yaml"), "w") as f:
            cfg = deepcopy(self.cfg)
            cfg.train_batch_size = batch_size
            cfg.random_seed = self.random_seed
            d = asdict(cfg) if is_dataclass(cfg) else dict(vars(cfg))
            d.update(train_batch_size=batch_size, random_seed=self.random_seed)
            yaml.safe_dump(json.loads(json.dumps(d, default=str)), f)
        self.log_writer = _summary_writer(logs_path)
        self.save_ckpt_per_i = save_ckpt_per_i
        self.log_per_i = log_per_i
        self.ema_alpha = ema_alpha
        opt = self.get_a_new_optimizer()
        self.opt, self.lr_scheduler = opt if isinstance(opt, (tuple, list)) else (opt, None)
        if load_ckpt_path:
            assert os.path.isfile(load_ckpt_path)
            self.train_i, self.train_epoch, _, self.ema_loss = self.load_ckpt(load_ckpt_path, True)
        else:
            if seed_init_train_params_if_no_ckpt_path:
                print("Train parameters are reinitialized with seed %s." % self.random_seed)
                self.reinit_train_parameters()
            self.train_i = self.train_epoch = self.ema_loss = 1
        self.other_train_init_final()

    # ---- base.py:194-225 ----------------------------------------------------------------------------------------------
    def train(self, total_epochs):
        self.best_ema_loss = float("inf")
        if getattr(self, "log_writer", None) is None:
            raise RuntimeError("Call `self.train_init()` to initialize training first!")
        print("Checkpoints dir: ", self.save_ckpt_dir)
        start_epoch = self.train_epoch
        self.set_train(True)
        for self.train_epoch in range(start_epoch, total_epochs + 1):
            for a_batch_samples, samp_n in self.data_generator:
                loss, log_dict = self.train_a_batch(a_batch_samples)
                self.ema_loss = self.ema_alpha * loss + (1 - self.ema_alpha) * self.ema_loss
                log_dict["Loss"] = loss
                log_dict["EMA Loss"] = self.ema_loss
                log_dict["Epoch"] = self.train_epoch
                if self.train_i % self.log_per_i == 0:
                    self.write_logs(self.train_i, log_dict)
                if self.ema_loss is not None and self.ema_loss < self.best_ema_loss:     # best-EMA checkpoint (:214-217)
                    print(f"New best ema_loss: {self.ema_loss:.4f} (previous {self.best_ema_loss:.4f})")
                    self.best_ema_loss = self.ema_loss
                    self.save_ckpt(self.train_i, self.train_epoch, loss, self.ema_loss)
                self.train_i += 1
        self.set_train(False)

    def write_logs(self, i, logs: dict):  # base.py:227-235
        for log_name, log in logs.items():
            if type(log) == dict:
                self.write_logs(i, {log_name + "-" + n: l for n, l in log.items()})
            else:
                self.log_writer.add_scalar(log_name, log, i)

    # ---- base.py:237-268 ----------------------------------------------------------------------------------------------
    def save_ckpt(self, i: int, epoch: int, loss: float, ema_loss: float = None):
        """One file named `Best` in self.save_ckpt_dir: {i, epoch, loss, ema_loss, train_modules{name: state_dict}, opt,
        lr_scheduler}."""
        train_modules = self.get_modules_for_training()
        ckpt = {
            "i": i,
            "epoch": epoch,
            "loss": loss,
            "ema_loss": ema_loss,
            "train_modules": {k: _cpu_state(v.state_dict()) for k, v in train_modules.items()},
            "opt": _cpu_state(self.opt.state_dict()) if getattr(self, "opt", None) is not None else None,
            "lr_scheduler": self.lr_scheduler.state_dict() if getattr(self, "lr_scheduler", None) is not None else None,
        }
        ckpt_path = os.path.join(self.save_ckpt_dir, "Best")
        torch.save(ckpt, ckpt_path)

    def load_ckpt(self, ckpt_path, restrict=True, load_opt=True):
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        train_modules = self.get_modules_for_training()
        for k in train_modules.keys():
            train_modules[k].load_state_dict(ckpt["train_modules"][k], restrict)
        if load_opt:
            self.opt.load_state_dict(ckpt["opt"])
            if getattr(self, "lr_scheduler", None) is not None and ckpt["lr_scheduler"] is not None:
                self.lr_scheduler.load_state_dict(ckpt["lr_scheduler"])
        print("Load %s checkpoint from %s." % (self.name_of_editor_and_model()[0], ckpt_path))
        return ckpt["i"], ckpt["epoch"], ckpt["loss"], ckpt["ema_loss"]


class HipAdamState(dict):
    """Adam moments of an editor's trainable tensors as plain device buffers -- {"t": step, "m": {key: tensor}, "v": {key: tensor}}
    -- with the optimizer interface the checkpoint code needs (`state_dict` / `load_state_dict`), readable and writable in the
    layout of `torch.optim.Adam.state_dict()`, which is what the `opt` entry of a reference `Best` checkpoint holds
    (R/editor/vllm_editors/base.py:237-255: `self.opt.state_dict()`).

    `torch_order`: [(key, element index or None)] -- the optimizer's parameters in torch's numbering (param_groups in order, the
    parameters of a group in order); an entry with an element index i is one scalar parameter stored as element i of a vector
    buffer (MEND_VL's `edit_lrs` ParameterList).  `group_sizes` / `group_lrs`: the param_groups.  Without `torch_order` only the
    private layout is understood.  A foreign `opt` that cannot be mapped (other layout, other shapes) is NOT an error: the state
    stays fresh (zero moments, step 0) and a warning says so -- training resumes from the checkpoint's modules."""

    def __init__(self, *a, torch_order=None, group_sizes=None, group_lrs=None, **k):
        super().__init__(*a, **k)
        self.torch_order, self.group_sizes, self.group_lrs = torch_order, group_sizes, group_lrs

    def state_dict(self):
        if not self.torch_order:
            return {"t": self["t"], "m": dict(self["m"]), "v": dict(self["v"])}
        state = {}
        if self["t"] > 0:
            for i, (key, el) in enumerate(self.torch_order):
                m, v = self["m"][key], self["v"][key]
                if el is not None:
                    m, v = m[el].clone(), v[el].clone()
                state[i] = {"step": torch.tensor(float(self["t"])), "exp_avg": m, "exp_avg_sq": v}
        groups, i0 = [], 0
        for n, lr in zip(self.group_sizes, self.group_lrs):
            groups.append({"lr": float(lr), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False,
                           "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                           "decoupled_weight_decay": False, "params": list(range(i0, i0 + n))})
            i0 += n
        return {"state": state, "param_groups": groups}

    def _fresh(self, why):
        import warnings
        warnings.warn("checkpoint 'opt' not loaded (%s): continuing with fresh Adam moments" % why)
        self["t"] = 0
        for mv in ("m", "v"):
            for t in self[mv].values():
                t.zero_()

    def load_state_dict(self, sd):
        if isinstance(sd, dict) and {"t", "m", "v"} <= set(sd):          # the private layout (round-2 checkpoints)
            if set(sd["m"]) != set(self["m"]):
                return self._fresh("moment keys differ")
            self["t"] = int(sd["t"])
            for mv in ("m", "v"):
                for k, v in sd[mv].items():
                    self[mv][k].copy_(v.to(self[mv][k].device))
            return
        if not (isinstance(sd, dict) and "state" in sd and "param_groups" in sd):
            return self._fresh("unknown layout")
        if not self.torch_order:
            return self._fresh("no parameter order known for this editor")
        n_params = sum(len(g["params"]) for g in sd["param_groups"])
        if n_params != len(self.torch_order):
            return self._fresh("%d parameters in the checkpoint, %d here" % (n_params, len(self.torch_order)))
        state = sd["state"]
        if len(state) == 0:                                              # saved before the first step
            self["t"] = 0
            return
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        steps = set()
        for i, (key, el) in zip(ids, self.torch_order):
            st = state.get(i)
            dst = self["m"][key] if el is None else self["m"][key][el]
            if st is None or tuple(st["exp_avg"].shape) != tuple(dst.shape):
                return self._fresh("state of parameter %d does not fit %s" % (i, key))
            steps.add(int(float(st["step"])))
        if len(steps) != 1:
            return self._fresh("parameters at different step counts")
        for i, (key, el) in zip(ids, self.torch_order):
            for mv, name in (("m", "exp_avg"), ("v", "exp_avg_sq")):
                dst = self[mv][key] if el is None else self[mv][key][el]
                dst.copy_(state[i][name].to(dst.device, dst.dtype))
        self["t"] = steps.pop()


def _cpu_state(sd):
    if isinstance(sd, torch.Tensor):
        return sd.detach().cpu()
    if isinstance(sd, dict):
        return {k: _cpu_state(v) for k, v in sd.items()}
    if isinstance(sd, (list, tuple)):
        return type(sd)(_cpu_state(v) for v in sd)
    return sd
