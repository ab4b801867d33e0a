"""One environment per subprocess, spoken to over a pipe with upstream's ``(cmd, data)`` messages
(cmd in {"step", "reset", "close"}; upstream worker.py).  ``Worker(env_config).child`` is the parent's end.

Environment failures are reported back through the pipe as ``WorkerException`` objects (upstream raises inside the
child, which leaves the parent's ``recv()`` hanging); ``environments.vec_env.PipeVecEnv`` is the consumer.
"""
import multiprocessing as mp
import traceback


class WorkerException(Exception):
    """Carries the text of an exception raised by an environment inside its worker process."""

    def __init__(self, ee, tb_text: str = ""):
        self.ee = ee
        self.tb = tb_text
        super().__init__(f"{type(ee).__name__}: {ee}\n{tb_text}")

    def re_raise(self):
        raise self.ee


class _EnvServer:
    """Child-side loop: owns the environment, answers one message at a time."""

    def __init__(self, pipe, env_config: dict, worker_id: int):
        from utils import create_env
        self.pipe = pipe
        self.env = create_env(env_config, worker_id=worker_id)

    def handle(self, cmd, data):
        if cmd == "step":
            return self.env.step(data)
        if cmd == "reset":
            return self.env.reset()
        if cmd == "close":
            return self.env.close()
        raise NotImplementedError(f"unknown worker command {cmd!r}")

    def serve(self):
        while True:
            try:
                cmd, data = self.pipe.recv()
            except (EOFError, KeyboardInterrupt):
                return
            try:
                reply = self.handle(cmd, data)
            except Exception as exc:  # hand the failure to the parent instead of dying silently
                reply = WorkerException(exc, traceback.format_exc())
            self.pipe.send(reply)
            if cmd == "close":
                self.pipe.close()
                return


def worker_process(remote, config: dict, worker_id: int = 0) -> None:
    _EnvServer(remote, config, worker_id).serve()


class Worker:
    """Handle of one environment process: ``child`` (pipe end) and ``process``."""

    def __init__(self, env_config: dict, worker_id: int = 0):
        self.child, remote = mp.Pipe()
        self.process = mp.Process(target=worker_process, args=(remote, env_config, worker_id), daemon=True)
        self.process.start()
