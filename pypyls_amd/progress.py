"""
Progress reporting of the resampling loops under ``verbose=True``.

The reference wraps its permutation / bootstrap loops in tqdm bars
(``utils.trange``, pyls/utils.py:128-152: "Running permutations", "Running
bootstraps", cleared when done).  Here the loops are asynchronous launches of
chunks of resamples, so a bar counts chunks the DEVICE has finished, not
chunks the host has queued: every chunk is followed by an event on the launch
stream and the bar advances when the event has completed (polled whenever the
host queues more work, and on a helper thread while the host waits in the
final sync).  The bar appears only when the leg is still running after
``DELAY_S`` seconds -- a 10 ms call prints nothing, the 200 s literal
``configs[3]`` call (10 000 permutations x 100 splits) shows where it is.
"""
import threading
import time

DELAY_S = 2.0


class Bar(object):
    """``total`` resamples; ``queued(n)`` after each asynchronous chunk, ``close()`` when the leg has been synced."""

    def __init__(self, desc, total, enabled, device=None):
        self.total, self.enabled = int(total), bool(enabled) and int(total) > 0
        self.events = []                               # (event, n) in launch order
        self.done = 0
        self.bar = None
        self.device = device
        self._lock = threading.Lock()
        self._stop = None
        if not self.enabled:
            return
        try:
            from tqdm import tqdm
            form = '{desc}: {percentage:3.0f}%|{bar}| {n_fmt}/{total_fmt} | {elapsed}<{remaining}'   # pyls/utils.py:146-147
            self.bar = tqdm(total=self.total, desc=desc, delay=DELAY_S, leave=False, ascii=True, bar_format=form)
        except Exception:                               # noqa: BLE001 -- no tqdm: the reference is silent then, too
            self.enabled = False                        # (pyls/utils.py:13-16)

    def queued(self, n):
        """n more resamples were just launched on the current stream of the bar's device."""
        if not self.enabled:
            return
        import torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        with self._lock:
            self.events.append((ev, int(n)))
        self.poll()

    def poll(self):
        if not self.enabled:
            return
        with self._lock:
            adv = 0
            while self.events and self.events[0][0].query():
                adv += self.events.pop(0)[1]
            if adv:
                self.done += adv
                self.bar.update(adv)
            elif self.bar is not None:
                self.bar.refresh()

    def watch(self, period=0.25):
        """Keep polling from a helper thread (the host is about to block in a device sync)."""
        if not self.enabled or self._stop is not None:
            return
        self._stop = threading.Event()

        def run():
            while not self._stop.wait(period):
                self.poll()
        threading.Thread(target=run, name='plsx-progress', daemon=True).start()

    def close(self):
        if not self.enabled:
            return
        if self._stop is not None:
            self._stop.set()
        with self._lock:
            self.events = []
            if self.bar is not None:
                self.bar.close()
        self.enabled = False
