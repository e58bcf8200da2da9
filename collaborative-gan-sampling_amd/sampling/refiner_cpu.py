"""Host-loop Refiner for 2-D data space (reference sampling/refiner_cpu.py:6-81).

``gan``/``sess`` are duck-typed exactly as the reference uses them:
``sess.run([gan.fake_sigmoid, gan.fake_saliency], feed_dict={gan.fake_samples: x})`` returns the
discriminator's sigmoid [B,1] and d mean_b softplus(-logit_b)/dx [B,2] in the order of the fetch
list; ``data.next_batch(B)`` supplies the real batch whose mean sigmoid is the loss baseline.
BASELINE config 1 ("CPU plumbing, no GPU") runs this class as is."""
import numpy as np

from .policy import PolicyAdaptive


class Refiner():
    def __init__(self, args):
        self.forward_steps = args.rollout_steps
        self.step_size = args.rollout_rate
        self.method = args.rollout_method
        self.policy = PolicyAdaptive(self.step_size, self.method)

    def set_env(self, gan, sess, data):
        self.sess, self.gan, self.data = sess, gan, data

    def _score(self, batch):
        sig, grad = self.sess.run([self.gan.fake_sigmoid, self.gan.fake_saliency],
                                  feed_dict={self.gan.fake_samples: batch})
        return sig, grad

    def manipulate_sample(self, fake_batch, mode='deterministic'):
        K, n = self.forward_steps, len(fake_batch)
        # loss baseline: mean D score of a fresh real batch (the fetch order here is the reference's :23)
        real = self.data.next_batch(fake_batch.shape[0])
        _, real_sigmoid = self.sess.run([self.gan.fake_saliency, self.gan.fake_sigmoid],
                                        feed_dict={self.gan.fake_samples: real})
        baseline = np.mean(real_sigmoid)

        x = fake_batch.copy()                       # never mutate the caller's batch
        sig, grad = self._score(x)
        loss = baseline - np.squeeze(sig)
        best_x, best_loss, best_step = x.copy(), loss.copy(), np.zeros_like(loss)
        traj = np.zeros((2, n, K + 1))              # x / y coordinates per step (2-D only, like the reference)
        traj[0, :, 0], traj[1, :, 0] = fake_batch[:, 0], fake_batch[:, 1]

        for i in range(K):
            self.policy.apply_gradient(x, grad, loss)           # in place on x
            sig, grad = self._score(x)
            loss = baseline - np.squeeze(sig)
            better = (best_loss - loss) > 0
            best_loss[better] = loss[better]
            best_x[better, :] = x[better, :]
            best_step[better] = i + 1
            traj[0, :, i + 1], traj[1, :, i + 1] = x[:, 0], x[:, 1]

        self.policy.reset_moving_average()
        self.optimal_step = best_step

        if mode == 'probabilistic':                 # one random step per sample, drawn per call; float64 out
            pick = np.random.randint(K + 1, size=n)
            rows = np.arange(n)
            return np.array([traj[0, rows, pick], traj[1, rows, pick]]).transpose()
        if mode == 'deterministic':
            return best_x
        raise NotImplementedError
