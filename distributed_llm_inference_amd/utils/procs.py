"""Process liveness that also sees zombies (an exited child that nobody has reaped yet still
answers kill(pid, 0))."""
import os


def pid_alive(pid: int) -> bool:
    if pid <= 0:
        return False
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:
        with open(f"/proc/{pid}/stat") as f:
            # pid (comm) state ...  — comm may contain spaces/parens: split after the last ')'
            state = f.read().rsplit(")", 1)[1].split()[0]
        return state not in ("Z", "X")
    except (FileNotFoundError, IndexError, OSError):
        return True
