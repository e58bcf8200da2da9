"""Host-loop Refiner for 2-D data space (behaviour of the reference's sampling/refiner_cpu.py:6-81).

``gan`` / ``sess`` are duck-typed the way the reference uses them:
``sess.run([gan.fake_sigmoid, gan.fake_saliency], feed_dict={gan.fake_samples: x})`` returns the discriminator's sigmoid
[B,1] and d mean_b softplus(-logit_b) / dx [B,2] in the order of the fetch list; ``data.next_batch(B)`` supplies the real
batch whose mean sigmoid is the loss baseline.  BASELINE config 1 ("CPU plumbing, no GPU") runs this class as is; the fused
on-device form of the same loop is cgs_amd.synthetic.Refiner.

Per call: one D evaluation of a real batch (baseline), K+1 evaluations of the moving batch, the policy update in place
between them; per sample the coordinates with the lowest loss (= highest D score) seen are kept.  'deterministic' returns
those (input dtype); 'probabilistic' returns, per sample, the coordinates after a uniformly drawn number of steps in
0..K (float64, drawn from the global numpy RNG once per call)."""
import numpy as np

from .policy import PolicyAdaptive

MODES = ("deterministic", "probabilistic")


class Refiner:
    def __init__(self, args):
        self.forward_steps = args.rollout_steps
        self.step_size = args.rollout_rate
        self.method = args.rollout_method
        self.policy = PolicyAdaptive(self.step_size, self.method)
        self.optimal_step = None

    def set_env(self, gan, sess, data):
        self.gan = gan
        self.sess = sess
        self.data = data

    # -- D evaluations -------------------------------------------------------------------------
    def _evaluate(self, points):
        g = self.gan
        sigmoid, saliency = self.sess.run([g.fake_sigmoid, g.fake_saliency], feed_dict={g.fake_samples: points})
        return np.squeeze(sigmoid), saliency

    def _real_baseline(self, batch_size):
        g = self.gan
        real = self.data.next_batch(batch_size)
        # (saliency first, sigmoid second: the fetch order of the reference's baseline call)
        _, sigmoid = self.sess.run([g.fake_saliency, g.fake_sigmoid], feed_dict={g.fake_samples: real})
        return np.mean(sigmoid)

    # -- the loop ------------------------------------------------------------------------------
    def manipulate_sample(self, fake_batch, mode='deterministic'):
        steps, count = self.forward_steps, len(fake_batch)
        baseline = self._real_baseline(fake_batch.shape[0])

        points = fake_batch.copy()                                  # the caller's batch is left untouched
        path = np.zeros((2, count, steps + 1))                      # (coordinate, sample, step): 2-D data only
        path[:, :, 0] = fake_batch[:, :2].T

        score, saliency = self._evaluate(points)
        loss = baseline - score
        keep_points, keep_loss = points.copy(), loss.copy()
        keep_step = np.zeros_like(loss)
        for k in range(1, steps + 1):
            self.policy.apply_gradient(points, saliency, loss)      # updates ``points`` in place
            score, saliency = self._evaluate(points)
            loss = baseline - score
            improved = (keep_loss - loss) > 0
            keep_loss[improved] = loss[improved]
            keep_points[improved, :] = points[improved, :]
            keep_step[improved] = k
            path[:, :, k] = points[:, :2].T
        self.policy.reset_moving_average()
        self.optimal_step = keep_step

        if mode not in MODES:
            raise NotImplementedError
        if mode == 'deterministic':
            return keep_points
        stop = np.random.randint(steps + 1, size=count)             # per-sample stopping step
        who = np.arange(count)
        return np.stack([path[0, who, stop], path[1, who, stop]], axis=1)
