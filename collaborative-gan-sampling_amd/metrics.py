"""Sample-quality metrics used right after the refinement path.

2-D mode metrics: the reference's sampling/utils_sampling.py:132-184 (mean distance to the nearest mode, rate of
"good" samples within ``thres`` of a mode, KL of the per-mode frequencies, JS of the frequencies including the
"no mode" bin), as reported in the reference's figures (README.md:26-28).
Image metric: the Frechet distance between two feature clouds (the formula behind the reference's tfgan
``mnist_frechet_distance``, nsgan/utils_mnist.py:85-116); the feature extractor is pluggable because the
reference's classifier graph is a missing binary blob.
"""
import numpy as np
from scipy import linalg, stats


def _mode_distances(samples, centeroids):
    samples, centeroids = np.asarray(samples), np.asarray(centeroids)
    return np.linalg.norm(samples[:, None, :] - centeroids[None, :, :], axis=2)          # [n, k]


def metrics_distance(samples, centeroids, thres):
    """(mean distance to the nearest mode, fraction of samples closer than ``thres`` to a mode).  utils_sampling.py:132-145."""
    dist_min = _mode_distances(samples, centeroids).min(axis=1)
    return float(np.mean(dist_min)), float((dist_min < thres).sum() / dist_min.size)


def freq_category(samples, centeroids, thres):
    """(per-mode frequencies among samples that hit a mode, frequencies incl. a last "no mode" bin).  :147-161."""
    hits = _mode_distances(samples, centeroids) < thres
    counts = hits.sum(axis=0)
    total, n = counts.sum(), hits.shape[0]
    freqs_valid = counts / total if total > 0 else np.ones(len(counts)) / len(counts)
    return freqs_valid, np.append(counts, n - total) / n


def kl_div(predictions, targets):
    """:169-177 (targets clipped away from 0/1; scipy entropy renormalises both)."""
    return float(stats.entropy(predictions, np.clip(targets, 1e-12, 1 - 1e-12)))


def metrics_diversity(real_batch, model_batch, centeroids, thres):
    """KL(model mode frequencies || real mode frequencies).  :163-167."""
    return kl_div(freq_category(model_batch, centeroids, thres)[0], freq_category(real_batch, centeroids, thres)[0])


def metrics_distribution(real_batch, model_batch, centeroids, thres):
    """JS divergence of the frequencies including the "no mode" bin.  :179-184."""
    fr, fm = freq_category(real_batch, centeroids, thres)[1], freq_category(model_batch, centeroids, thres)[1]
    avg = 0.5 * (fr + fm)
    return 0.5 * kl_div(fr, avg) + 0.5 * kl_div(fm, avg)


def frechet_distance(feat_real, feat_fake):
    """|mu_r - mu_f|^2 + Tr(S_r + S_f - 2 (S_r S_f)^{1/2}) between two [n, d] feature clouds."""
    fr, ff = np.asarray(feat_real, dtype=np.float64), np.asarray(feat_fake, dtype=np.float64)
    mu_r, mu_f = fr.mean(0), ff.mean(0)
    s_r, s_f = np.cov(fr, rowvar=False), np.cov(ff, rowvar=False)
    covmean, _ = linalg.sqrtm(s_r.dot(s_f), disp=False)
    covmean = covmean.real if np.iscomplexobj(covmean) else covmean
    return float(((mu_r - mu_f) ** 2).sum() + np.trace(s_r) + np.trace(s_f) - 2.0 * np.trace(covmean))
