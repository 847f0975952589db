"""numpy's legacy RandomState(42) stream, replayed from its raw MT19937 words.

Every k-means fit of the reference restarts from `random_state=42` (clustering.py:211-218, 751-752), so all of
its random draws are functions of ONE fixed sequence of 32-bit Mersenne-Twister outputs.  The words are generated
once per process (by numpy itself) and the three draw kinds the path needs are replayed from them, vectorised:

* `randint(0, n, size)`  -- numpy's masked rejection sampling: candidates `word & mask` (mask = 2^b - 1 >= n - 1),
  one word each, kept when <= n - 1 (`_bounded_integers.pyx`, legacy `use_masked=True`);
* `random_sample()`      -- `((a >> 5) * 2^26 + (b >> 6)) / 2^53` from two consecutive words;
* `uniform(size=N)`      -- `0.0 + 1.0 * random_sample()`: the same doubles (computed on the device from the
  resident words by `rhccq_mt_uniforms`, so that neither the host nor PCIe sees them).

tests/test_cabi_cpu.py::test_mt_replay_equals_numpy_randomstate pins the replay against RandomState itself."""
import threading

import numpy as np

SEED = 42


class MtWords:
    def __init__(self, seed=SEED):
        self._rs = np.random.RandomState(seed)
        self._lock = threading.Lock()                      # replays may run on several host threads
        self.words = np.zeros(0, np.uint32)

    def ensure(self, n):
        """raw words [0, n) as a uint32 array (grown geometrically; the generator object keeps its position)."""
        if len(self.words) < n:
            with self._lock:
                if len(self.words) < n:
                    grow = max(n - len(self.words), len(self.words), 1 << 20)
                    # full-range uint32 draws are the raw 32-bit outputs, one word each (rng == 0xFFFFFFFF branch)
                    more = self._rs.randint(0, 1 << 32, size=grow, dtype=np.uint32)
                    self.words = np.concatenate([self.words, more])
        return self.words

    def randint(self, pos, n, size):
        """RandomState.randint(0, n, size) starting at word `pos`: (values int64[size], words consumed)."""
        rng = n - 1
        if size == 0:
            return np.zeros(0, np.int64), 0
        if rng == 0:
            return np.zeros(size, np.int64), 0                      # numpy draws nothing for a one-value range
        if rng >= 0xFFFFFFFF:
            raise ValueError("ranges of 2^32 and more use numpy's 64-bit path, which this replay does not model")
        mask = np.uint32((1 << int(rng).bit_length()) - 1)
        accept_rate = n / (int(mask) + 1)
        win = int(size / accept_rate * 1.05) + 256
        while True:
            w = self.ensure(pos + win)[pos:pos + win] & mask
            hit = np.flatnonzero(w <= rng)
            if len(hit) >= size:
                hit = hit[:size]
                return w[hit].astype(np.int64), int(hit[-1]) + 1
            win *= 2

    def double(self, pos):
        """random_sample() at word `pos` (consumes two words)."""
        w = self.ensure(pos + 2)
        return (float(int(w[pos]) >> 5) * 67108864.0 + float(int(w[pos + 1]) >> 6)) / 9007199254740992.0

    def doubles(self, pos, count):
        """uniform(size=count) at word `pos` on the host (the device computes the same from the resident words)."""
        w = self.ensure(pos + 2 * count)[pos:pos + 2 * count]
        return ((w[0::2] >> np.uint32(5)).astype(np.float64) * 67108864.0 + (w[1::2] >> np.uint32(6)).astype(np.float64)) / 9007199254740992.0
