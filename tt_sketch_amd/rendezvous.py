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
    """``broadcast`` / ``allgather`` / ``barrier`` of byte strings between the ranks of one node.

    Every launch runs under a fresh SESSION id agreed on by a handshake, and every file of the launch carries it in
    its name -- a directory that still holds the files of an earlier launch (one that crashed before ``close()``, or a
    fixed ``TTSK_RDV_DIR`` reused by a long-lived driver) cannot feed a stale payload (an old NCCL id, an old gather
    piece) to a new rank:

      1. rank r writes ``hello.r`` with a random token of its own (fresh per process);
      2. rank 0 collects the tokens and publishes ``session`` = (its own random id, the token list);
      3. rank r accepts a session only if position r of its token list is r's own token (a session built from a stale
         ``hello.r`` is not) and answers ``ack.r`` = the session id; rank 0 re-reads the hello files and publishes again
         until every ack names the current session.

    The directory is created 0700 and must belong to the calling user.  ``close()`` removes this rank's files and
    the last rank out removes the directory."""

    def __init__(self, rank: int, world: int, directory: Optional[str] = None, timeout: float = 300.0):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if directory is None:
            directory = os.environ.get("TTSK_RDV_DIR")
        if directory is None:
            key = "_".join((os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                            _parent_key()))
            directory = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ttsk_rdv_{os.getuid()}_{key}")
        self.dir = directory
        os.makedirs(self.dir, mode=0o700, exist_ok=True)
        st = os.stat(self.dir)
        if st.st_uid != os.getuid():
            raise PermissionError(f"rendezvous directory {self.dir} belongs to another user")
        if st.st_mode & 0o077:
            os.chmod(self.dir, 0o700)
        self._seq = 0
        self._mine: List[str] = []
        self.session = self._handshake()

    # -- files
    def _write(self, name: str, payload: bytes) -> str:
        path = os.path.join(self.dir, name)
        tmp = path + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.rename(tmp, path)
        if path not in self._mine:
            self._mine.append(path)
        return path

    def _read(self, name: str) -> Optional[bytes]:
        try:
            with open(os.path.join(self.dir, name), "rb") as f:
                return f.read()
        except FileNotFoundError:
            return None

    def _poll(self, what: str, step):
        """call step() until it returns something other than None"""
        deadline = time.monotonic() + self.timeout
        delay = 1e-4
        while True:
            out = step()
            if out is not None:
                return out
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: {what} (directory {self.dir})")
            time.sleep(delay)
            delay = min(delay * 2, 0.01)

    def _handshake(self) -> str:
        token = os.urandom(8).hex()
        self._write(f"hello.{self.rank}", token.encode())
        if self.rank == 0:
            sid = os.urandom(8).hex()
            published: List[Optional[str]] = [None]

            def step():
                tokens = [self._read(f"hello.{r}") for r in range(self.world)]
                if any(t is None for t in tokens):
                    return None
                line = sid + " " + " ".join(t.decode() for t in tokens)
                if line != published[0]:                      # first time, or a hello file was replaced by a fresh rank
                    self._write("session", line.encode())
                    published[0] = line
                # an ack counts if it names THIS session and the acknowledging rank's current token (the line may have
                # been republished since because ANOTHER rank's stale hello was replaced)
                def good(r):
                    a = self._read(f"ack.{r}")
                    parts = a.decode().split() if a is not None else []
                    return len(parts) == self.world + 1 and parts[0] == sid and parts[1 + r] == tokens[r].decode()
                return sid if all(good(r) for r in range(1, self.world)) else None
            return self._poll("the ranks never acknowledged the session", step)

        def step():
            raw = self._read("session")
            if raw is None:
                return None
            parts = raw.decode().split()
            if len(parts) != self.world + 1 or parts[1 + self.rank] != token:
                return None                                   # a session of an earlier launch, or built from my stale hello
            self._write(f"ack.{self.rank}", raw)
            return parts[0]
        return self._poll("rank 0 never published a session naming this rank", step)

    def _path(self, tag: str, rank: int) -> str:
        return os.path.join(self.dir, f"{self.session}_{tag}.{rank}")

    def _put(self, tag: str, payload: bytes) -> None:
        self._write(f"{self.session}_{tag}.{self.rank}", payload)

    def _get(self, tag: str, rank: int) -> bytes:
        name = f"{self.session}_{tag}.{rank}"
        return self._poll(f"rank {rank} never wrote {name}", lambda: self._read(name))

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
        """Every rank has read everything it will ever read once it passes the closing barrier; each rank then removes
        its own files, says ``bye`` and the last one out removes the directory."""
        try:
            self.barrier()
        except TimeoutError:
            pass
        closing = self._path(f"{self._seq:06d}_gather", self.rank)
        for p in self._mine:
            if p != closing:                  # the closing barrier's own file may still be polled by slower ranks
                try:
                    os.remove(p)
                except OSError:
                    pass
        self._mine = []
        try:
            with open(os.path.join(self.dir, f"{self.session}_bye.{self.rank}"), "wb"):
                pass
            if all(os.path.exists(os.path.join(self.dir, f"{self.session}_bye.{r}")) for r in range(self.world)):
                for name in os.listdir(self.dir):             # everybody is past the barrier: nothing is read any more
                    try:
                        os.remove(os.path.join(self.dir, name))
                    except OSError:
                        pass
                os.rmdir(self.dir)
        except OSError:
            pass
