"""Dataset base classes (R/dataset/__init__.py:118-126)."""
from abc import ABC, abstractmethod


class BaseEditData(ABC):
    def __init__(self, data) -> None:
        super().__init__()
        self.data = data

    @abstractmethod
    def dataset_name(self):
        """return dataset name"""
        raise


# ------------------------------------------------------------------------------------------------------------------
# ParallelDataset: batches prepared ahead of the training step (R/dataset/__init__.py:13-114).
#
# Same constructor, iteration protocol and -- for a given seed -- the same sequence of id batches as the reference's
# class: one rng draws the batch size and reshuffles at every wrap-around; without `drop_last` the tail of an epoch is
# completed with the head of the next permutation; an `iter()` pass ends once `sample_count` samples were yielded while
# the buffer keeps whatever was prepared beyond that (tests/golden/parallel_dataset_ids.json holds id sequences
# captured from the reference's class).
#
# What is MI355X-specific: the reference's producer thread runs `get_data_by_ids_func` -- model forwards included -- on a
# SECOND GPU holding a second model copy (R/utils/__init__.py:149-156).  One MI355X has room and idle CUs for both, so
# here the producer thread runs on the SAME device under its own HIP stream (`device=`): every item carries an event
# recorded on that stream; `__next__` makes the consumer's current stream wait for it and marks the item's tensors as
# used by the consumer stream, so the caching allocator cannot hand their memory back to the producer too early.
# ------------------------------------------------------------------------------------------------------------------
import atexit
import queue
import threading

import numpy as np


def _walk_tensors(obj, fn):
    import torch
    if isinstance(obj, torch.Tensor):
        fn(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _walk_tensors(v, fn)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _walk_tensors(v, fn)


class ParallelDataset:
    def __init__(self, sample_count: int, get_data_by_ids_func, batch_size=256, shuffle=True, buffer_size=64, drop_last=False,
                 random_seed=None, return_samp_n=True, device=None) -> None:
        self.sample_count = sample_count
        self.set_batch_size(batch_size)
        self.shuffle, self.drop_last, self.return_samp_n = shuffle, drop_last, return_samp_n
        self.rng = np.random.default_rng(random_seed)
        self.buffer_size = max(1, int(buffer_size))
        self.now_yield_i = 0
        self._get = get_data_by_ids_func
        self._device = device
        self._side = None
        if device is not None and str(device).startswith("cuda"):
            import torch
            self._side = torch.cuda.Stream(device=device)
            self._side.wait_stream(torch.cuda.current_stream(device))      # weights / caches queued so far are complete for the producer
        self._ids = self._id_batches()
        self._q = queue.Queue(maxsize=self.buffer_size)
        self._stop = threading.Event()
        self._err = None
        self._worker = threading.Thread(target=self._produce, daemon=True)
        self._worker.start()
        atexit.register(self.close)

    def set_batch_size(self, batch_size):
        if type(batch_size) != list and type(batch_size) != int:
            raise
        if type(batch_size) == list and len(batch_size) == 0:
            raise
        if type(batch_size) == int and batch_size <= 0:
            raise
        batch_size = [batch_size] if type(batch_size) != list else batch_size
        self.batch_size = np.array([min(bs, self.sample_count) for bs in batch_size])

    def _id_batches(self):
        """The endless stream of id batches (R/dataset/__init__.py:62-82)."""
        n = self.sample_count
        order = np.arange(n)
        if self.shuffle:
            self.rng.shuffle(order)
        cur = 0
        while True:
            bs = int(self.rng.choice(self.batch_size))
            end = cur + bs
            ids = order[cur:end]
            if end < n:
                cur = end
            else:                       # wrap: new permutation, the tail is completed from its head
                order = np.arange(n)
                if self.shuffle:
                    self.rng.shuffle(order)
                if end > n and self.drop_last:
                    cur = 0
                    continue
                cur = end - n
                ids = np.concatenate([ids, order[:cur]], 0)
            yield ids

    def _produce(self):
        try:
            import contextlib
            ctx = contextlib.nullcontext()
            if self._side is not None:
                import torch
                torch.cuda.set_device(self._device)
                ctx = torch.cuda.stream(self._side)
            with ctx:
                for ids in self._ids:
                    if self._stop.is_set():
                        return
                    d = self._get(ids)
                    ev = None
                    if self._side is not None:
                        import torch
                        ev = torch.cuda.Event()
                        ev.record(self._side)
                    item = (d, len(ids), ev)
                    while not self._stop.is_set():
                        try:
                            self._q.put(item, timeout=0.2)
                            break
                        except queue.Full:
                            continue
        except BaseException as e:          # surfaced by the consumer; a silent dead producer would hang the training loop
            self._err = e

    def close(self):
        """Stop the producer and wait for it: it must not be inside a HIP call when the interpreter shuts down."""
        self._stop.set()
        w = getattr(self, "_worker", None)
        if w is not None and w.is_alive() and w is not threading.current_thread():
            w.join(timeout=120)

    def __del__(self):
        if hasattr(self, "_stop"):
            self._stop.set()

    def __len__(self):
        if len(self.batch_size) > 1:
            print("The number of data batches is not accurate since `batch_size` is a list")
        bs = self.batch_size.mean()
        return int(np.floor(self.sample_count / bs)) if self.drop_last else int(np.ceil(self.sample_count / bs))

    def __iter__(self):
        self.now_yield_i = 0
        return self

    def __next__(self):
        if self.now_yield_i >= self.sample_count:
            raise StopIteration
        while True:
            if self._err is not None:
                raise RuntimeError("ParallelDataset producer failed") from self._err
            try:
                d, data_n, ev = self._q.get(timeout=0.2)
                break
            except queue.Empty:
                continue
        if ev is not None:
            import torch
            cur = torch.cuda.current_stream(self._device)
            cur.wait_event(ev)
            _walk_tensors(d, lambda t: t.record_stream(cur) if t.is_cuda else None)
        self.now_yield_i += data_n
        return (d, data_n) if self.return_samp_n else d
