"""The 2-D toy data of the reference's synthetic/ package (synthetic/Datasets.py:9-13,16-80): same constructor
arguments, attributes (``centeroids``, ``std``) and global-numpy-RNG consumption order, because
``refiner_cpu.Refiner.manipulate_sample`` draws its real batch from it (refiner_cpu.py:22)."""
import numpy as np


class NoiseDataset:
    def __init__(self, distr='Gaussian', dim=2, var=1):
        self.distr, self.dim, self.var = distr, dim, var

    def next_batch(self, batch_size=64):
        if self.distr != 'Gaussian':
            raise NotImplementedError
        return np.random.randn(batch_size, self.dim)


class ToyDataset:
    def __init__(self, distr='8Gaussians', scale=2, ratio=0.5):
        self.distr, self.scale, self.ratio = distr, scale, ratio
        if distr in ('8Gaussians', 'Imbal-8Gaussians'):
            d = 1. / np.sqrt(2)
            ring = [(1, 0), (d, d), (d, -d), (-1, 0), (0, 1), (0, -1), (-d, d), (-d, -d)]
            self.centers = ring
            self.centeroids = np.array(ring) * scale / 1.414
            self.std = 0.02 * scale / 1.414
        elif distr == '25Gaussians':
            self.centers = [(x, y) for x in range(-2, 3) for y in range(-2, 3)]
            self.centeroids = np.array(self.centers) * scale
            self.std = 0.05

    def _with_noise(self, parts, n_random, n_modes, batch_size):
        if n_random > 0:
            parts.append(self.centeroids[np.random.randint(n_modes, size=n_random), :])
        return np.concatenate(parts) + np.random.normal(0.0, self.std, size=(batch_size, 2))

    def next_batch(self, batch_size=64):
        if self.distr == 'Imbal-8Gaussians':
            n_major = int(batch_size * self.ratio / 2)
            assert n_major > 0
            n_minor = int((batch_size - n_major * 2) / 6)
            assert n_minor > 0
            parts = [np.repeat(self.centeroids[:2, :], n_major, axis=0), np.repeat(self.centeroids[2:, :], n_minor, axis=0)]
            return self._with_noise(parts, batch_size - n_major * 2 - n_minor * 6, 8, batch_size)
        if self.distr in ('8Gaussians', '25Gaussians'):
            k = len(self.centeroids)
            rep = int(batch_size / k)
            return self._with_noise([np.repeat(self.centeroids, rep, axis=0)], batch_size - rep * k, k, batch_size)
        raise NotImplementedError(self.distr)
