"""Memory-mapped safetensors reader with per-tensor selection (+ a small writer).

The reference makes every process load the whole checkpoint with
``AutoModelForCausalLM.from_pretrained`` and then drop (orchestration.py:39-55) or simply keep
(Worker1.py:60-75) what it does not need.  A pipeline stage here opens the shard files with
``mmap`` and touches only the byte ranges of the tensors that belong to its layer range, so
host memory and load time scale with the stage, not the model.

File format: 8-byte little-endian header length N, N bytes of JSON
``{name: {"dtype","shape","data_offsets":[begin,end]}, "__metadata__": {...}}``, then the
raw little-endian tensor bytes; offsets are relative to the end of the header.
"""
from __future__ import annotations

import json
import mmap
import os
import struct
import warnings
from typing import Dict, Iterable, List, Optional, Tuple

import torch

_DTYPES = {
    "F64": torch.float64, "F32": torch.float32, "F16": torch.float16, "BF16": torch.bfloat16,
    "I64": torch.int64, "I32": torch.int32, "I16": torch.int16, "I8": torch.int8, "U8": torch.uint8,
    "BOOL": torch.bool, "F8_E4M3": torch.float8_e4m3fn, "F8_E5M2": torch.float8_e5m2,
}
_DTYPE_NAMES = {v: k for k, v in _DTYPES.items()}


class SafetensorsFile:
    """One ``.safetensors`` file, mapped read-only. Tensors are zero-copy views of the map."""

    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        head = self._f.read(8)
        if len(head) != 8:
            raise ValueError(f"{path}: truncated safetensors header")
        (n,) = struct.unpack("<Q", head)
        size = os.fstat(self._f.fileno()).st_size
        if n > size - 8:
            raise ValueError(f"{path}: header length {n} exceeds file size {size}")
        meta = json.loads(self._f.read(n).decode("utf-8"))
        self.metadata = meta.pop("__metadata__", {}) or {}
        self.entries: Dict[str, dict] = meta
        self._data_start = 8 + n
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ) if size > 0 else None
        for name, e in self.entries.items():
            b, en = e["data_offsets"]
            if not (0 <= b <= en <= size - self._data_start):
                raise ValueError(f"{path}: tensor {name!r} has offsets outside the file")

    def keys(self) -> List[str]:
        return list(self.entries)

    def __contains__(self, name: str) -> bool:
        return name in self.entries

    def shape(self, name: str) -> Tuple[int, ...]:
        return tuple(self.entries[name]["shape"])

    def get(self, name: str) -> torch.Tensor:
        e = self.entries[name]
        dt = _DTYPES[e["dtype"]]
        b, en = e["data_offsets"]
        shape = tuple(e["shape"])
        numel = 1
        for s in shape:
            numel *= s
        if numel == 0:
            return torch.empty(shape, dtype=dt)
        buf = memoryview(self._mm)[self._data_start + b:self._data_start + en]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)      # read-only mmap: callers copy/cast, never write
            t = torch.frombuffer(buf, dtype=dt, count=numel)
        return t.reshape(shape)

    def close(self):
        # Views handed out by get() keep the mmap alive through the buffer protocol;
        # closing is best-effort and only releases our own handles.
        try:
            if self._mm is not None:
                self._mm.close()
        except BufferError:
            pass
        self._f.close()


class CheckpointReader:
    """A Hugging Face checkpoint directory: single-file or sharded safetensors, or a
    ``pytorch_model.bin`` (loaded with ``weights_only=True``)."""

    def __init__(self, model_dir: str):
        self.model_dir = model_dir
        self._files: Dict[str, SafetensorsFile] = {}
        self._where: Dict[str, str] = {}
        self._bin: Optional[Dict[str, torch.Tensor]] = None
        idx = os.path.join(model_dir, "model.safetensors.index.json")
        single = os.path.join(model_dir, "model.safetensors")
        if os.path.isfile(idx):
            with open(idx) as f:
                self._where = dict(json.load(f)["weight_map"])
        elif os.path.isfile(single):
            sf = self._open("model.safetensors")
            self._where = {k: "model.safetensors" for k in sf.keys()}
        else:
            sts = sorted(p for p in os.listdir(model_dir) if p.endswith(".safetensors"))
            if sts:
                for p in sts:
                    for k in self._open(p).keys():
                        self._where[k] = p
            else:
                binp = os.path.join(model_dir, "pytorch_model.bin")
                if not os.path.isfile(binp):
                    raise FileNotFoundError(f"no safetensors or pytorch_model.bin under {model_dir}")
                self._bin = torch.load(binp, map_location="cpu", weights_only=True, mmap=True)
                self._where = {k: "pytorch_model.bin" for k in self._bin}

    def _open(self, fname: str) -> SafetensorsFile:
        sf = self._files.get(fname)
        if sf is None:
            sf = SafetensorsFile(os.path.join(self.model_dir, fname))
            self._files[fname] = sf
        return sf

    def keys(self) -> List[str]:
        return list(self._where)

    def __contains__(self, name: str) -> bool:
        return name in self._where

    def get(self, name: str) -> torch.Tensor:
        if self._bin is not None:
            return self._bin[name]
        return self._open(self._where[name]).get(name)

    def first_present(self, names: Iterable[str]) -> Optional[str]:
        for n in names:
            if n in self._where:
                return n
        return None

    @property
    def files_opened(self) -> List[str]:
        return sorted(self._files)

    def close(self):
        for f in self._files.values():
            f.close()


def save_safetensors(tensors: Dict[str, torch.Tensor], path: str, metadata: Optional[Dict[str, str]] = None):
    """Write a safetensors file (used by tests and by `tools` that export packed stages)."""
    header: Dict[str, dict] = {}
    off = 0
    blobs = []
    for name in sorted(tensors):
        t = tensors[name].detach().contiguous().cpu()
        raw = t.view(torch.uint8).reshape(-1).numpy().tobytes() if t.numel() else b""
        header[name] = {"dtype": _DTYPE_NAMES[t.dtype], "shape": list(t.shape),
                        "data_offsets": [off, off + len(raw)]}
        off += len(raw)
        blobs.append(raw)
    if metadata:
        header["__metadata__"] = metadata
    hj = json.dumps(header, separators=(",", ":")).encode("utf-8")
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for b in blobs:
            f.write(b)
