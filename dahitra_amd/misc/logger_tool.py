"""The two host utilities the training / evaluation harness needs around the step:

Logger   a tee: every message goes to stdout and is appended to a log file (the reference's `log.txt` / `log_test.txt`,
         models/trainer.py:72-75, models/evaluator.py:46-49).  The file is opened per message, so a crashed run keeps its log.
Timer    progress bookkeeping for the "imps / est" figures of the batch log line (models/trainer.py:136-142): from the
         fraction of work done it extrapolates the hours that remain; a separate stage clock measures throughput.

The reference's whole public surface is kept (import names: misc/logger_tool.py), including the Timer's context-manager form
and its `str_estimated_*` helpers, which none of the reference's own scripts call."""
import sys
import time

_BANNER = "================ (%s) ================\n"


class Logger:
    def __init__(self, outfile):
        self.log_path = outfile
        self.terminal = sys.stdout
        self.write(_BANNER % time.strftime("%c"))

    def write(self, message):
        self.terminal.write(message)
        with open(self.log_path, "a") as fh:
            fh.write(message)

    def _write_items(self, d, fmt):
        self.write("".join(fmt % kv for kv in d.items()))

    def write_dict(self, d):                # numeric values, fixed 7 decimals (score tables)
        self._write_items(d, "%s: %.7f ")

    def write_dict_str(self, d):            # anything printable (the args namespace)
        self._write_items(d, "%s: %s ")

    def flush(self):
        self.terminal.flush()


class Timer:
    """`est_remaining` (seconds) and `estimated_remaining()` (hours) follow the last `update_progress(fraction)`;
    `get_stage_elapsed()` / `lapse()` read the stage clock, which `lapse()` and `reset_stage()` restart."""

    def __init__(self, starting_msg=None):
        now = time.time()
        self.start = now
        self.stage_start = now
        self.est_remaining = 0.0
        if starting_msg is not None:
            print(starting_msg, time.ctime(now))

    # `with Timer(...) as t:` is accepted (misc/logger_tool.py:41-45: the context manager does nothing on either side)
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return None

    def update_progress(self, progress):
        self.elapsed = spent = time.time() - self.start
        self.est_total = spent / progress if progress > 0 else float("inf")
        self.est_remaining = self.est_total - spent
        self.est_finish = int(self.start + self.est_total) if progress > 0 else None

    def estimated_remaining(self):
        return self.est_remaining / 3600

    def str_estimated_remaining(self):
        return "%sh" % self.estimated_remaining()

    def str_estimated_complete(self):
        if getattr(self, "est_finish", None) is None:
            raise AttributeError("Timer.str_estimated_complete() before update_progress()")     # as the reference: no est_finish yet
        return time.ctime(self.est_finish)

    def get_stage_elapsed(self):
        return time.time() - self.stage_start

    def reset_stage(self):
        self.stage_start = time.time()

    def lapse(self):
        now = time.time()
        elapsed, self.stage_start = now - self.stage_start, now
        return elapsed
