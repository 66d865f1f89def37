"""The two host utilities the training / evaluation harness needs around the step:

Logger   a tee: every message goes to stdout and is appended to a log file (the reference's `log.txt` / `log_test.txt`,
         models/trainer.py:72-75, models/evaluator.py:46-49).  The file is opened per message, so a crashed run keeps its log.
Timer    progress bookkeeping for the "imps / est" figures of the batch log line (models/trainer.py:136-142): from the
         fraction of work done it extrapolates the hours that remain; a separate stage clock measures throughput.

Only what `CDTrainer`, `CDEvaluator` and the reference's scripts call is provided (import names: misc/logger_tool.py)."""
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

    def update_progress(self, progress):
        spent = time.time() - self.start
        self.est_remaining = spent / progress - spent if progress > 0 else float("inf")

    def estimated_remaining(self):
        return self.est_remaining / 3600

    def get_stage_elapsed(self):
        return time.time() - self.stage_start

    def reset_stage(self):
        self.stage_start = time.time()

    def lapse(self):
        now = time.time()
        elapsed, self.stage_start = now - self.stage_start, now
        return elapsed
