"""Environment subprocess with the upstream pipe protocol (worker.py): messages are ``(cmd, data)`` with
cmd in {"step", "reset", "close"}; ``Worker(env_config).child`` is the parent end of the pipe."""
import multiprocessing
import multiprocessing.connection
import sys
import traceback


def worker_process(remote: multiprocessing.connection.Connection, config: dict, worker_id: int = 0) -> None:
    from utils import create_env
    env = create_env(config, worker_id=worker_id)
    handlers = {"step": lambda d: env.step(d), "reset": lambda d: env.reset(), "close": lambda d: env.close()}
    while True:
        try:
            cmd, data = remote.recv()
            if cmd not in handlers:
                raise NotImplementedError(cmd)
            remote.send(handlers[cmd](data))
            if cmd == "close":
                remote.close()
                return
        except EOFError:
            return
        except Exception as exc:  # surface env failures to the parent instead of hanging its recv()
            raise WorkerException(exc)


class Worker:
    """One environment in one process."""
    child: multiprocessing.connection.Connection
    process: multiprocessing.Process

    def __init__(self, env_config: dict, worker_id: int = 0):
        self.child, parent = multiprocessing.Pipe()
        self.process = multiprocessing.Process(target=worker_process, args=(parent, env_config, worker_id), daemon=True)
        self.process.start()


class WorkerException(Exception):
    """Raised inside the worker process; carries the formatted traceback of the environment error."""

    def __init__(self, ee):
        self.ee = ee
        self.tb = "".join(traceback.format_exception(*sys.exc_info()))
        super().__init__(f"{ee}\n{self.tb}")

    def re_raise(self):
        raise self.ee
