"""Refiner: collaborative refinement of a G activation map (reference sampling/collaborator.py:7-88).

Same constructor, ``set_env`` / ``set_constraints`` / ``compute_forward_logits_and_grad`` /
``build_refiner`` and the same readable attributes (``default_logit``, ``optimal_logit``,
``optimal_step``, ``optimal_feature``).  The reference's ``build_refiner`` emits a K-times unrolled
TF graph that a later ``sess.run`` executes; here it executes: it returns the refined images.

Execution paths (both all-HIP, no CPU fallback; ``refiner.path`` says which one the last call took and
``refiner.why_generic`` why the engine was not taken):
  * engine:  if ``discriminator`` / ``feature_to_data`` are a ``cgs_amd.model.GAN``'s own methods -- bound, or wrapped
             in ``functools.partial`` the way nsgan/GAN.py:174-175 wraps them -- and ``func_loss`` computes the
             cross entropy against ones (GAN.loss_refine, or any closure that evaluates to softplus(-logit), :176-177),
             the whole loop runs as the fused device program (``engine.RefineEngine``): state resident in HBM, no host
             sync, the K-step program replayed as a hipGraph (``use_graph``, default on; eager fallback in-process).
             The engine differentiates softplus(-logit) analytically: a ``func_loss`` with a CUSTOM backward or side
             effects that merely evaluates to BCE-vs-ones must set ``refiner.force_generic = True``.
  * generic: any other callables built from ``cgs_amd.ops``; the loop below differentiates them with
             ``torch.autograd`` (each op's backward-data is a HIP kernel) and applies the fused
             update / select kernels.  Nothing is fused across layers; the first fall to this path warns once with the
             reason.  With ``use_graph`` (default) the K-step loop of a (shape, K, mode, rate, clip) is CAPTURED into a
             hipGraph during its second call and replayed from then on (round 6): the reference's ``build_refiner`` is
             itself a graph BUILDER -- it calls ``discriminator`` / ``feature_to_data`` / ``func_loss`` K+1 times while the TF graph
             is built and ``sess.run`` replays that graph (sampling/collaborator.py:41-88, nsgan/GAN.py:182-183,270) -- so
             "the callables' Python runs once, the recorded device program runs per batch" is the reference's own contract.
             ``ops.bn``'s moving-average update is part of the recorded program; a capture that is refused (a callable that
             synchronises with the host, say) falls back to eager launches in the same process and ``graph_fallback`` says why.

``refiner.logical_batch = b`` (extension): ``build_refiner(feature[G*b], ...)`` is G reference calls at the reference's own
batch size b (nsgan/main.py:32: 64) in ONE launch per layer -- D's batch statistics per logical batch (nsgan/GAN.py:175), the
probabilistic step draw per logical batch (collaborator.py:54-56), the optimizer reset per call (:86) -- instead of G
launch-latency-bound calls.  On the generic path the G batches run one after the other (same results).
"""
import functools
import weakref

import numpy as np
import torch

from .policy import PolicyAdaptive
from .. import kernels as K
from .. import lib as L


class Refiner():
    def __init__(self, rollout_steps, rollout_rate, rollout_method="momentum"):
        self.forward_steps = rollout_steps
        self.optimizer = PolicyAdaptive(rollout_rate, rollout_method)
        self.log = False
        self.vmin = None
        self.vmax = None
        self.use_graph = True           # engine path: the K-step program is captured into a hipGraph at the first call of a
                                        # (batch, K, rate, mode) and replayed afterwards; a refused capture falls back to eager
                                        # launches in the same process (``graph_fallback`` then holds the reason)
        self.graph_fallback = None
        self.path = None                # "engine" | "generic": which execution form the last build_refiner took
        self.contraction = "f32"        # engine path: "f32" = exact fp32 matrix instructions (the reference's precision); "bx6" opts the big
                                        # layers into the split-bf16 contraction (include/cgs_hip.h, cgs_set_contraction)
        self.indices_batch = None       # last probabilistic draw
        self.logical_batch = None       # b: build_refiner's rows are consecutive batches of b samples, each with the reference's per-batch
                                        # semantics (batch statistics, index draw); None = the rows are ONE batch (the reference's call)
        self._generic_graphs = {}       # generic path: (shape, K, mode, rate, method, clip) -> _GenericGraph (captured loop + its static buffers)
        self._generic_seen = set()      # ... keys that already ran once eagerly (the capture happens at the SECOND call of a key)
        self._gstream = None            # ... the side stream the loops are warmed up and captured on (packed-weight workspaces are per stream)
        self.force_generic = False      # True: never take the engine (e.g. a func_loss with a custom backward that evaluates to BCE-vs-ones)
        self.why_generic = None         # why the last engine detection said no (None while the engine is taken)

    def set_env(self, discriminator, feature_to_data, func_loss):
        if (getattr(self, "discriminator", None) is not discriminator or getattr(self, "feature_to_data", None) is not feature_to_data
                or getattr(self, "func_loss", None) is not func_loss):
            # other callables: the generic path's recorded programs belong to the old ones
            self._generic_graphs.clear()
            self._generic_seen.clear()
        self.discriminator = discriminator
        self.feature_to_data = feature_to_data
        self.func_loss = func_loss

    def set_constraints(self, vmin, vmax):
        self.vmin = vmin
        self.vmax = vmax
        print("set_constraints: self.vmin = {:.2f}, self.vmax = {:.2f}".format(self.vmin, self.vmax))

    # -- one forward/backward evaluation (collaborator.py:26-39) ---------------------------------
    def compute_forward_logits_and_grad(self, current_feature, need_grad=True):
        feature = current_feature.detach().requires_grad_(need_grad)
        with torch.set_grad_enabled(need_grad):
            forward_logits = self.discriminator(self.feature_to_data(feature))
            forward_grad = None
            if need_grad:
                forward_loss = self.func_loss(forward_logits)
                forward_grad = torch.autograd.grad(forward_loss.sum(), feature)[0]   # tf.gradients sums ys
        flat = forward_logits.detach().reshape(forward_logits.shape[0], -1).contiguous()
        return K.bce_ones_grad_rowmean(flat)[1], forward_grad                        # per-sample mean logit (:34-37)

    # -- engine detection ------------------------------------------------------------------------
    @staticmethod
    def _unwrap(fn, allowed):
        """A bound method, or a ``functools.partial`` of one with keyword arguments only (nsgan/GAN.py:174-175 hands the refiner
        ``partial(self.discriminator, is_training=True, reuse=True)``) -> (owner, plain function, keywords); None otherwise."""
        kw = {}
        while isinstance(fn, functools.partial):
            if fn.args:
                return None
            kw = dict(fn.keywords or {}, **kw)
            fn = fn.func
        owner, func = getattr(fn, "__self__", None), getattr(fn, "__func__", None)
        if owner is None or func is None or not set(kw) <= set(allowed):
            return None
        return owner, func, kw

    _BCE_PROBE = {}            # key of func_loss -> (weak reference to it, verdict); an entry dies with its function / its object
    _WARNED = set()            # generic-path reasons already warned about (once per process and reason)

    @classmethod
    def _loss_is_bce_ones(cls, func_loss, device):
        """Is ``func_loss`` the unreduced cross entropy against all-ones labels, softplus(-logit) (nsgan/GAN.py:176-177)?  The
        reference passes a local closure, so identity cannot tell: the function is evaluated ONCE on 40 logits -- 24 evenly spaced
        over [-40, 40] (a clipped / saturating variant deviates there) and 16 seeded random ones -- and compared with softplus(-l)
        to 4 fp32 ulp relative + 1e-7 (shape kept = no reduction).  Anything else -- another loss, a reduction, an exception --
        keeps the generic path, which differentiates whatever it is.  Only VALUES are compared: see ``force_generic``."""
        from ..model import GAN
        if func_loss is GAN.loss_refine:
            return True
        # a bound method is a fresh object at every attribute access: key it by (object, function), and follow the object with WeakMethod
        bound = hasattr(func_loss, "__self__") and hasattr(func_loss, "__func__")
        key = ("m", id(func_loss.__self__), id(func_loss.__func__)) if bound else ("f", id(func_loss))
        hit = cls._BCE_PROBE.get(key)
        if hit is not None:
            alive = hit[0]()
            if alive is not None and (alive == func_loss if bound else alive is func_loss):
                return hit[1]
        ok = False
        try:
            l = np.concatenate([np.linspace(-40.0, 40.0, 24), np.random.RandomState(2019).normal(0.0, 6.0, 16)]).astype(np.float32).reshape(40, 1)
            with torch.no_grad():
                got = func_loss(torch.from_numpy(l).to(device))
            want = np.logaddexp(0.0, -l.astype(np.float64))
            ok = (torch.is_tensor(got) and tuple(got.shape) == (40, 1)
                  and bool((np.abs(got.detach().double().cpu().numpy() - want) <= 4 * 2.0 ** -23 * np.abs(want) + 1e-7).all()))
        except Exception:                                   # noqa: BLE001 (a loss that cannot take a [40, 1] tensor is not this one)
            ok = False
        try:       # (the reference makes a new closure per model build: keep no function alive, drop the verdict with it)
            drop = lambda _r, key=key: cls._BCE_PROBE.pop(key, None)        # noqa: E731
            cls._BCE_PROBE[key] = ((weakref.WeakMethod if bound else weakref.ref)(func_loss, drop), ok)
        except TypeError:                                   # not weak-referenceable (a builtin): probe it again next time
            pass
        return ok

    def _engine_owner(self):
        """The ``model.GAN`` whose layer lists ARE the three callables of ``set_env`` -- in any of the spellings the reference's
        wiring allows (the bound methods, ``functools.partial`` of them, a BCE-vs-ones closure) -- or None, with the reason left
        in ``why_generic``."""
        from ..model import GAN
        self.why_generic = None

        def no(reason):
            self.why_generic = reason
            return None
        if self.force_generic:
            return no("refiner.force_generic is set")
        d = self._unwrap(self.discriminator, ("is_training", "reuse"))
        f = self._unwrap(self.feature_to_data, ("is_training",))
        if d is None:
            return no("discriminator is not a bound GAN method (or a keyword-only functools.partial of one): a lambda / free function hides the layer list")
        if f is None:
            return no("feature_to_data is not a bound GAN method (or a keyword-only functools.partial of one)")
        owner = d[0]
        if not isinstance(owner, GAN) or f[0] is not owner:
            return no("discriminator and feature_to_data do not belong to one cgs_amd.model.GAN")
        if d[1] is GAN.discriminator_refine:
            if d[2]:
                return no("discriminator_refine takes no keywords")
        elif d[1] is GAN.discriminator:
            if d[2].get("is_training", True) is not True:   # (GAN.discriminator's default; the engine's D runs batch-statistics bn)
                return no("discriminator is bound with is_training=False: the engine's D runs batch-statistics bn (nsgan/GAN.py:175)")
        else:
            return no("discriminator is not GAN.discriminator / GAN.discriminator_refine")
        if f[1] is not GAN.feature_to_data or f[2].get("is_training", False) is not False:
            return no("feature_to_data is not GAN.feature_to_data in inference mode")
        if self.optimizer.method not in ("momentum", "sgd"):
            return no(f"rollout_method {self.optimizer.method!r}: the engine runs momentum / sgd")
        if not self._loss_is_bce_ones(self.func_loss, owner.device):
            return no("func_loss does not evaluate to the unreduced cross entropy against ones, softplus(-logit) (nsgan/GAN.py:176-177)")
        return owner

    def _engine_for(self, batch):
        owner = self._engine_owner()
        return None if owner is None else owner.engine(batch, use_graph=self.use_graph, contraction=self.contraction)

    def _groups(self, B):
        """Number of logical batches in a call of B rows (``logical_batch``)."""
        b = self.logical_batch
        if not b or int(b) == B:
            return 1
        if int(b) <= 0 or B % int(b):
            raise L.CgsError(f"logical_batch={b} does not divide the {B} rows handed to build_refiner")
        return B // int(b)

    def build_refiner(self, fake_feature, real_batch, mode='deterministic', indices=None):
        """collaborator.py:41-88.  ``real_batch`` only feeds statistics the reference computes and never
        uses (:44-45); it is accepted and ignored.  ``indices`` (extension) replays a fixed probabilistic
        draw; by default one is drawn per call with np.random.randint(K+1, size=B) (:54-56) -- per logical batch,
        in order, under ``logical_batch``."""
        K_steps = self.forward_steps
        B = fake_feature.shape[0]
        G = self._groups(B)
        if mode == 'probabilistic':
            if indices is not None:
                self.indices_batch = np.asarray(indices)
            else:       # G consecutive reference calls draw G times from the global stream
                self.indices_batch = np.concatenate([np.random.randint(K_steps + 1, size=B // G) for _ in range(G)])
        elif mode != 'deterministic':
            raise NotImplementedError

        owner = self._engine_owner()
        if owner is not None:
            self.path = "engine"
            args = (fake_feature, K_steps, self.optimizer.lambda_, self.optimizer.method, mode,
                    self.indices_batch if mode == 'probabilistic' else None, self.vmin, self.vmax)
            eng = owner.engine(B, use_graph=self.use_graph, contraction=self.contraction, bn_groups=G)
            try:
                out = eng.refine(*args)
            except L.GraphCaptureError as ex:                     # the capture block refused (HIP, allocator, another thread's HIP call)
                # same process, the SAME engine and buffers, launched kernel by kernel instead; the refiner stays eager from here on
                # and says why.  (Argument errors, OOM or a failure of the eager warm-up are not caught: they are what they are.)
                self.graph_fallback = str(ex)
                self.use_graph = False
                try:
                    torch.cuda.synchronize(fake_feature.device)
                except Exception:                                 # noqa: BLE001
                    pass
                eng = owner.demote_engine_to_eager(B, self.contraction, G)      # (the one place the cache's key layout lives: model.GAN)
                out = eng.refine(*args)
            img, d_l, o_l, o_s, o_f = out
            # the engine returns its own (cached, reused) buffers: hand out copies, so that a second build_refiner -- the
            # reference builds a deterministic and a probabilistic refiner side by side, nsgan/GAN.py:182-183 -- does not
            # overwrite the first one's results in place
            self.default_logit, self.optimal_logit = d_l.clone(), o_l.clone()
            self.optimal_step, self.optimal_feature = o_s.clone(), o_f.clone()
            self.optimizer.reset_moving_average()
            return img.clone()

        # ---- generic path -----------------------------------------------------------------------
        self.path = "generic"
        if self.why_generic not in self._WARNED:
            self._WARNED.add(self.why_generic)
            import warnings
            warnings.warn("cgs_amd Refiner: this wiring runs on the generic (ops + autograd) path, several times slower than the fused "
                          f"engine at small batches -- {self.why_generic}", RuntimeWarning, stacklevel=2)
        if G == 1:
            return self._run_generic(fake_feature, mode, self.indices_batch if mode == 'probabilistic' else None)
        b = B // G
        outs, attrs = [], {k: [] for k in ("default_logit", "optimal_logit", "optimal_step", "optimal_feature")}
        for j in range(G):                                    # G reference calls, one after the other
            idx = self.indices_batch[j * b:(j + 1) * b] if mode == 'probabilistic' else None
            outs.append(self._run_generic(fake_feature[j * b:(j + 1) * b], mode, idx))
            for k in attrs:
                attrs[k].append(getattr(self, k))
        for k, v in attrs.items():
            setattr(self, k, torch.cat(v))
        return torch.cat(outs)

    # -- generic path: eager, or captured into a hipGraph and replayed ------------------------------
    _ATTRS = ("default_logit", "optimal_logit", "optimal_step", "optimal_feature")

    def _run_generic(self, fake_feature, mode, indices):
        """One reference call on the generic path.  First call of a (shape, K, mode, rate, method, clip): eager (creates variables,
        packs weights, sizes workspaces).  Second call: eager once more on the refiner's side stream (so the per-stream packed-weight
        workspaces exist there), then the same loop CAPTURED on that stream; this and every later call replay the hipGraph.  The
        graph is dropped when a weight it reads moves on (in-place update, ``ops.set_variables``, ``K.WS.invalidate``)."""
        dev = fake_feature.device
        if not (self.use_graph and dev.type == "cuda"):
            return self._build_generic(fake_feature, mode, indices)
        key = (tuple(fake_feature.shape), dev.index, self.forward_steps, mode, float(self.optimizer.lambda_), self.optimizer.method,
               self.vmin, self.vmax)
        gg = self._generic_graphs.get(key)
        if gg is not None and not gg.valid():
            del self._generic_graphs[key]               # a weight moved on: the recorded program reads stale packed copies
            gg = None
        if gg is None:
            if key not in self._generic_seen:
                self._generic_seen.add(key)
                return self._build_generic(fake_feature, mode, indices)
            try:
                gg = _GenericGraph(self, fake_feature, mode)
            except L.GraphCaptureError as ex:
                self.graph_fallback = str(ex)
                self.use_graph = False
                try:
                    torch.cuda.synchronize(dev)
                except Exception:                                 # noqa: BLE001
                    pass
                return self._build_generic(fake_feature, mode, indices)
            self._generic_graphs[key] = gg
        return gg.run(self, fake_feature, indices)

    def _build_generic(self, fake_feature, mode, indices, forced=None):
        """collaborator.py:41-88 launched kernel by kernel.  ``forced`` (captured form): the device buffer the probabilistic step
        indices are read from (filled before every replay) instead of a fresh upload of ``indices``."""
        K_steps = self.forward_steps
        self.current_feature = fake_feature.detach().clone().contiguous()
        self.current_logit, self.forward_grad = self.compute_forward_logits_and_grad(self.current_feature)
        self.default_logit = self.current_logit
        self.optimal_feature = self.current_feature.clone()
        self.optimal_logit = self.current_logit.clone().contiguous()
        self.optimal_step = torch.ones_like(self.optimal_logit)
        if mode == 'probabilistic' and forced is None:
            forced = torch.as_tensor(indices, dtype=torch.int32).to(fake_feature.device)

        for i in range(K_steps):
            self.current_feature = self.optimizer.apply_gradient(self.current_feature, self.forward_grad)
            if self.vmin and self.vmax:                       # the reference's truthiness test (:69)
                self.current_feature = K.clip(self.current_feature, self.vmin, self.vmax)
            self.current_logit, self.forward_grad = self.compute_forward_logits_and_grad(
                self.current_feature, need_grad=(i + 1 < K_steps))     # the K-th gradient is dead code in the reference graph
            K.refine_select(self.current_feature, self.current_logit.contiguous(), forced, i,
                            self.optimal_feature, self.optimal_logit, self.optimal_step)

        self.optimizer.reset_moving_average()
        with torch.no_grad():
            return self.feature_to_data(self.optimal_feature)


class _GenericGraph:
    """The generic path's K-step loop of one call signature as a hipGraph: static input / index buffers, the loop's tensors in the
    graph's private pool, the results copied out after every replay."""

    def __init__(self, refiner, feature, mode):
        import gc
        dev = feature.device
        self.dev = dev
        self.feature = torch.empty_like(feature)                       # static input: filled before every replay
        self.forced = torch.zeros(feature.shape[0], dtype=torch.int32, device=dev) if mode == 'probabilistic' else None
        if refiner._gstream is None:
            refiner._gstream = torch.cuda.Stream(dev)
        st = refiner._gstream
        cur = torch.cuda.current_stream(dev)
        self.feature.copy_(feature)
        st.wait_stream(cur)
        # eager pass on the capture stream: packs the weights into THIS stream's workspaces, recording which weight tensors the
        # callables read (their versions are what ``valid`` watches); its results are discarded (the replay below recomputes them)
        K.WS.record = []
        try:
            with torch.cuda.stream(st):
                refiner._build_generic(self.feature, mode, None, forced=self.forced)
            torch.cuda.synchronize(dev)
            self.weights = [(r, r()._version) for r in {id(r()): r for r in K.WS.record if r() is not None}.values()]
        finally:
            K.WS.record = None
        from .. import ops
        gc_was_on = gc.isenabled()
        gc.disable()                     # (no finalizer that reaches HIP inside a capture: engine.RefineEngine.refine has the story)
        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=st, capture_error_mode="thread_local"):
                img = refiner._build_generic(self.feature, mode, None, forced=self.forced)
            self.out = (img,) + tuple(getattr(refiner, k) for k in Refiner._ATTRS)
        except L.CgsError:
            raise
        except Exception as ex:          # noqa: BLE001 (HIP / allocator / a callable that synchronises refused the capture)
            raise L.GraphCaptureError(f"{type(ex).__name__}: {str(ex)[:300]}") from ex
        finally:
            if gc_was_on:
                gc.enable()
        cur.wait_stream(st)
        # the capture pass bumped no versions on the device, but in-place ops recorded in it (ops.bn's moving averages) bumped the
        # tensors' version counters on the host: the state to compare with is the one AFTER the capture
        self.weights = [(r, r()._version) for r, _ in self.weights]
        self.stamp = (K.WS.epoch, K.WS.clears, ops.generation())

    def valid(self):
        from .. import ops
        if self.stamp != (K.WS.epoch, K.WS.clears, ops.generation()):
            return False
        for r, ver in self.weights:
            w = r()
            if w is None or w._version != ver:
                return False
        return True

    def run(self, refiner, feature, indices):
        self.feature.copy_(feature)
        if self.forced is not None:
            self.forced.copy_(torch.as_tensor(np.asarray(indices), dtype=torch.int32))
        self.graph.replay()
        # the graph's own tensors are overwritten by the next replay: hand out copies (as the engine path does)
        img, *attrs = [t.clone() for t in self.out]
        for k, v in zip(Refiner._ATTRS, attrs):
            setattr(refiner, k, v)
        refiner.optimizer.reset_moving_average()
        return img
