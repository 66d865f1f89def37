"""Logger / Timer of the reference's misc/logger_tool.py: a tee to stdout + an append-mode log file, and the
progress / ETA bookkeeping CDTrainer._timer_update reads (models/trainer.py:136-142)."""
import sys
import time


class Logger:
    def __init__(self, outfile):
        self.terminal = sys.stdout
        self.log_path = outfile
        self.write('================ (%s) ================\n' % time.strftime("%c"))

    def write(self, message):
        self.terminal.write(message)
        with open(self.log_path, mode='a') as f:
            f.write(message)

    def write_dict(self, d):
        self.write(''.join('%s: %.7f ' % (k, v) for k, v in d.items()))

    def write_dict_str(self, d):
        self.write(''.join('%s: %s ' % (k, v) for k, v in d.items()))

    def flush(self):
        self.terminal.flush()


class Timer:
    def __init__(self, starting_msg=None):
        self.start = self.stage_start = time.time()
        self.elapsed = self.est_total = self.est_remaining = 0.0
        self.est_finish = int(self.start)
        if starting_msg is not None:
            print(starting_msg, time.ctime(self.start))

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        return

    def update_progress(self, progress):
        self.elapsed = time.time() - self.start
        self.est_total = self.elapsed / progress
        self.est_remaining = self.est_total - self.elapsed
        self.est_finish = int(self.start + self.est_total)

    def str_estimated_complete(self):
        return str(time.ctime(self.est_finish))

    def str_estimated_remaining(self):
        return str(self.est_remaining / 3600) + 'h'

    def estimated_remaining(self):
        return self.est_remaining / 3600

    def get_stage_elapsed(self):
        return time.time() - self.stage_start

    def reset_stage(self):
        self.stage_start = time.time()

    def lapse(self):
        now = time.time()
        out, self.stage_start = now - self.stage_start, now
        return out
