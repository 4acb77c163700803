"""Host-side rendezvous of the ranks of ONE node without any framework: a directory of small files.

The only thing the multi-GPU sketch path needs from the host before RCCL is up is to carry rank 0's
128-byte communicator id to the other ranks (``RcclComm``); afterwards barriers and reductions are
RCCL calls on the device.  ``torch.distributed`` would do, but importing torch into this process also
loads torch's own bundled ROCm runtime (``torch/lib/libamdhip64.so``, ``librccl.so``,
``libhsa-runtime64.so``) next to the system copies ``libttsk.so`` links against -- two HIP runtimes
and two RCCLs in one process, which is what ended round 1's only N = 2 run in
``double free or corruption`` at interpreter exit.

Ranks launched by ``python -m torch.distributed.run`` (the driver's launcher) share a parent process
and the ``MASTER_PORT`` / ``TORCHELASTIC_RUN_ID`` environment; that names the directory.  Files are
written under a temporary name and renamed, so a reader never sees a partial payload.
"""
from __future__ import annotations

import os
import time
from typing import List, Optional


def _parent_key() -> str:
    ppid = os.getppid()
    start = ""
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]      # starttime: unique per launch even if pids recycle
    except (OSError, IndexError):
        pass
    return f"{ppid}_{start}"


class FileRendezvous:
    """``broadcast`` / ``allgather`` / ``barrier`` of byte strings between the ranks of one node."""

    def __init__(self, rank: int, world: int, directory: Optional[str] = None, timeout: float = 300.0):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if directory is None:
            directory = os.environ.get("TTSK_RDV_DIR")
        if directory is None:
            key = "_".join((os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                            _parent_key()))
            directory = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ttsk_rdv_{key}")
        self.dir = directory
        os.makedirs(self.dir, exist_ok=True)
        self._seq = 0
        self._mine: List[str] = []

    def _path(self, tag: str, rank: int) -> str:
        return os.path.join(self.dir, f"{tag}.{rank}")

    def _put(self, tag: str, payload: bytes) -> None:
        path = self._path(tag, self.rank)
        tmp = path + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.rename(tmp, path)
        self._mine.append(path)

    def _get(self, tag: str, rank: int) -> bytes:
        path = self._path(tag, rank)
        deadline = time.monotonic() + self.timeout
        delay = 1e-4
        while True:
            try:
                with open(path, "rb") as f:
                    return f.read()
            except FileNotFoundError:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: rank {rank} never wrote {path}")
                time.sleep(delay)
                delay = min(delay * 2, 0.01)

    def _tag(self, name: str) -> str:
        self._seq += 1
        return f"{self._seq:06d}_{name}"

    def broadcast(self, payload: Optional[bytes], root: int = 0) -> bytes:
        tag = self._tag("bcast")
        if self.rank == root:
            if payload is None:
                raise ValueError("the root rank must supply the payload")
            self._put(tag, bytes(payload))
            return bytes(payload)
        return self._get(tag, root)

    def allgather(self, payload: bytes) -> List[bytes]:
        tag = self._tag("gather")
        self._put(tag, bytes(payload))
        return [self._get(tag, r) for r in range(self.world)]

    def barrier(self) -> None:
        self.allgather(b"")

    def close(self) -> None:
        """Every rank has read everything it will ever read once it passes this barrier."""
        try:
            self.barrier()
            last = self._mine[-1] if self._mine else None
            for p in self._mine:
                if p != last:                 # the closing barrier's own files may still be polled by slower ranks
                    try:
                        os.remove(p)
                    except OSError:
                        pass
        except TimeoutError:
            pass
        self._mine = []
