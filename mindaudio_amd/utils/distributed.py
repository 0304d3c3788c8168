"""Mirror of mindaudio/utils/distributed.py:4-29 (pure host logic: an index permutation per epoch)."""
import numpy as np


class DistributedSampler:
    """Yields dataset indices for one epoch.  As in the reference the seed is incremented *before* each epoch's
    permutation and the global NumPy generator is re-seeded with it (distributed.py:17-20); `group=True` keeps every
    group_size-th index starting at `rank` (distributed.py:24-25)."""

    def __init__(self, dataset, rank, group_size, shuffle=True, seed=0, group=True):
        self.dataset_len = len(dataset)
        self.rank, self.group_size = rank, group_size
        self.shuffle, self.seed, self.group = shuffle, seed, group

    def __len__(self):
        return self.dataset_len

    def __iter__(self):
        order = np.arange(self.dataset_len)
        if self.shuffle:
            self.seed = (self.seed + 1) & 0xFFFFFFFF
            np.random.seed(self.seed)
            order = np.random.permutation(self.dataset_len)
        picked = order[self.rank::self.group_size] if self.group else order
        return iter(picked)
