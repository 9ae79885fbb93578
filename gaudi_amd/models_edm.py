"""Host mirror of models_edm.get_model / EnVariationalDiffusion's sampling interface
(models_edm.py:61-104, edm/equivariant_diffusion/en_diffusion.py:958-1067), backed by libgaudi_hip.so."""
from __future__ import annotations

import numpy as np

from . import checkpoint
from ._lib import GaudiError
from .engine import Engine


def _to_numpy(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def _like_ref(a):
    """The reference returns torch tensors; do the same when torch is importable."""
    try:
        import torch
        return torch.from_numpy(np.ascontiguousarray(a))
    except Exception:  # pragma: no cover
        return a


class DistributionRings:
    """models_edm.DistributionRings (models_edm.py:21-58): ring-count sampler over the dataset histogram
    (utils/helpers.py:64-95; shipped as data in gaudi_amd/data/ring_tables.json, histogram order preserved).
    sample() draws through torch.distributions.Categorical exactly as the reference does, so a run seeded with
    torch.manual_seed reproduces the reference's ring counts."""

    def __init__(self, dataset="cata"):
        import torch
        from torch.distributions.categorical import Categorical

        from .analyze import ring_tables
        hist = ring_tables()["n_nodes"]["cata" if dataset == "peri" else dataset]
        self.n_nodes = torch.tensor([int(k) for k in hist])
        self.keys = {int(k): i for i, k in enumerate(hist)}
        prob = np.array([hist[k] for k in hist])
        prob = prob / np.sum(prob)
        self.prob = torch.from_numpy(prob).float()
        self.m = Categorical(torch.tensor(prob))

    def sample(self, n_samples=1):
        return self.n_nodes[self.m.sample((n_samples,))]

    def log_prob(self, batch_n_nodes):
        import torch
        assert len(batch_n_nodes.size()) == 1
        idcs = torch.tensor([self.keys[i.item()] for i in batch_n_nodes])
        return torch.log(self.prob + 1e-30)[idcs]


class PropertyNorm:
    """DistributionProperty(only_norm=True): mean/std + (un)normalize (models_edm.py:107-113,186-192).
    mean/std are NOT stored in checkpoints (they come from the dataset) and must be supplied."""

    def __init__(self, mean, std):
        self.mean = np.asarray(_to_numpy(mean), np.float32)
        self.std = np.asarray(_to_numpy(std), np.float32)

    def _ms(self, like):
        """mean / std in the type of ``like``: the reference's DistributionProperty holds torch tensors, and a target closure
        calls prop_dist.unnormalize on a torch prediction that requires grad (generation_guidance.py:205-211)."""
        if hasattr(like, "detach"):
            import torch
            return (torch.as_tensor(self.mean, dtype=like.dtype, device=like.device),
                    torch.as_tensor(self.std, dtype=like.dtype, device=like.device))
        return self.mean, self.std

    def normalize(self, pred):
        mean, std = self._ms(pred)
        return (pred - mean) / std

    def unnormalize(self, pred):
        mean, std = self._ms(pred)
        return pred * std + mean


class LinearTarget:
    """Declarative target function T(pred) = w . pred + c of the predictor outputs -- the native form of the
    closures in generation_guidance.py:198-211.  Calling it evaluates T on the GPU (predictor forward)."""

    def __init__(self, cond_predictor, weights, const=0.0, name="linear"):
        self.cond_predictor = cond_predictor
        self.weights = np.asarray(weights, np.float32)
        self.const = float(const)
        self.name = name

    def __call__(self, _input, _node_mask, _edge_mask, _t):
        pred = _to_numpy(self.cond_predictor(_input, _node_mask, _edge_mask, _t))
        return _like_ref((pred @ self.weights + self.const).astype(np.float32))


class ZTarget:
    """A target closure over (z, node_mask, edge_mask, t) that depends on z BOTH through the predictor and directly -- the
    reference differentiates any function of z_s (en_diffusion.py:899-903).  Per reverse step the host evaluates the closure on
    the device's z_s with the predictor output replaced by a leaf holding the device's prediction: torch.autograd then yields
    dT/dpred (fed to the GPU reverse pass) and the direct dT/dz (added to its result before the clip): gaudi_sample_cbz."""

    def __init__(self, cond_predictor, target, name="closure(z)"):
        self.cond_predictor = cond_predictor
        self.target = target
        self.name = name

    def grad(self, z, pred, t, node_mask, edge_mask):
        import torch
        cp = self.cond_predictor
        zt = torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32)).requires_grad_(True)
        p = torch.from_numpy(np.ascontiguousarray(pred, dtype=np.float32)).requires_grad_(True)
        B = p.shape[0]
        cp._override, cp._override_used = p, False
        try:
            with torch.enable_grad():
                val = self.target(zt, torch.from_numpy(node_mask), torch.from_numpy(edge_mask), torch.full((B, 1), float(t)))
                gp, gz = torch.autograd.grad(val.sum(), (p, zt), allow_unused=True)
        finally:
            cp._override = None
        gp = np.zeros_like(pred) if gp is None else gp.detach().numpy().astype(np.float32)
        gz = np.zeros_like(z) if gz is None else gz.detach().numpy().astype(np.float32)
        return gp, gz

    def __call__(self, _input, _node_mask, _edge_mask, _t):
        return self.target(_input, _node_mask, _edge_mask, _t)


class PredTarget:
    """Arbitrary differentiable target T = fn(pred, t) of the predictor outputs (any closure the reference's
    sample_guidance accepts has this form, generation_guidance.py:187-211).  ``fn`` takes a torch tensor pred [B,K]
    (requires_grad) and the step time t and returns one value per molecule [B]; only its K-vector gradient is taken on
    the host (torch.autograd on a [B,K] leaf) -- the predictor forward and reverse passes stay on the GPU
    (gaudi_sample_cb)."""

    def __init__(self, cond_predictor, fn, name="custom"):
        self.cond_predictor = cond_predictor
        self.fn = fn
        self.name = name

    def grad(self, pred: np.ndarray, t: float) -> np.ndarray:
        import torch

        p = torch.from_numpy(np.ascontiguousarray(pred, dtype=np.float32)).requires_grad_(True)
        with torch.enable_grad():
            val = self.fn(p, t)
            (g,) = torch.autograd.grad(val.sum(), p, allow_unused=True)
        return np.zeros_like(pred) if g is None else g.detach().numpy().astype(np.float32)

    def __call__(self, _input, _node_mask, _edge_mask, _t):
        import torch

        pred = torch.from_numpy(_to_numpy(self.cond_predictor(_input, _node_mask, _edge_mask, _t)))
        t = float(np.asarray(_to_numpy(_t)).reshape(-1)[0]) if np.ndim(_to_numpy(_t)) else float(_t)
        return self.fn(pred, t)


class _DirectZ(Exception):
    pass


# autograd nodes a function that is affine in its leaf can be built from (name without the "BackwardN" suffix).  Anything else
# -- clamp, where, abs, threshold, relu, pow, exp ... -- is "not affine", whatever the numerical probes say: the backward pass of
# a mask-style kink (clamp / where / masked_fill) carries no grad_fn either, and random probes only see the side of the kink
# they land on.  Products / quotients of two pred-dependent factors pass this list and are caught by the
# differentiable-gradient check below.
_AFFINE_NODES = frozenset((
    "Neg", "Mul", "Add", "Sub", "Rsub", "Div", "Select", "Slice", "Index", "IndexSelect", "Gather", "Narrow", "Sum", "Mean",
    "View", "Reshape", "UnsafeView", "Squeeze", "Unsqueeze", "Expand", "Repeat", "T", "Transpose", "Permute", "Clone",
    "Contiguous", "Cat", "Stack", "Unbind", "Split", "SplitWithSizes", "Chunk", "ToCopy", "Alias", "Copy", "CopySlices",
    "AsStrided", "Mm", "Mv", "Dot", "Addmm", "Matmul", "Bmm", "Linear", "AccumulateGrad", "Flatten", "Unflatten", "Movedim",
    "Diagonal", "Flip", "Roll", "Addcmul"))


def _graph_is_affine(val) -> bool:
    """Every node of val's autograd graph is one a linear map is made of (no kinks, no transcendental functions)."""
    seen, todo = set(), [val.grad_fn]
    while todo:
        node = todo.pop()
        if node is None or node in seen:
            continue
        seen.add(node)
        name = type(node).__name__
        cut = name.find("Backward")
        if (name[:cut] if cut >= 0 else name) not in _AFFINE_NODES:
            return False
        todo.extend(fn_ for fn_, _ in node.next_functions)
    return True


def _affine_probe(fn, p, t, ref):
    """One probe of affine_target_weights: -> dT/dpred of row 0 (== ref in every row when ref is given), or None."""
    import torch
    with torch.enable_grad():
        val = fn(p, t)
        if not torch.is_tensor(val) or not val.requires_grad or not _graph_is_affine(val):
            return None
        (g,) = torch.autograd.grad(val.sum(), p, create_graph=True, allow_unused=True)
    if g is None or g.requires_grad:
        return None  # no dependence on pred at all / the gradient depends on pred: not affine
    g = g.detach()
    if not bool(torch.isfinite(g).all()):
        return None
    if ref is None:
        ref = g[0].clone()
    return ref if bool((g == ref[None]).all()) else None


def affine_target_weights(fn, K: int, T: int, B: int = 3):
    """-> w [K] if fn(pred [B,K], t) -> [B] is  w . pred + c  with the same non-zero w for every molecule and every t, else
    None.  Three independent checks, all of which must hold: (i) the autograd graph of the value consists of linear nodes only
    (_AFFINE_NODES: a clamp / where / abs is refused by name, wherever its kink lies); (ii) dT/dpred does not depend on pred --
    asked for a differentiable gradient, it carries no grad_fn (all Hessian-vector products vanish identically); (iii)
    numerically: bit-equal gradients at random predictions of five very different magnitudes, at the first reverse step's
    t = 1, the last step's t = 1/T and one in between, and in every row of the batch.  w = 0 is 'not affine' too: the fused
    kernel would run an unguided chain where the callback path follows the closure faithfully (e.g. a closure that is
    switched off outside a window of t).  The probe visits three values of t; affine_gradient_holds() visits ALL of the
    chain's (GaudiModel._run runs it beside the device call).  Anything the probe cannot digest is 'not affine' (the general
    callback path, which is always correct)."""
    import torch
    gen = torch.Generator().manual_seed(1234)  # (torch's default generator keys the noise: not touched)
    ref = None
    try:
        for scale_p in (1.0, 1e-2, 30.0, 1e3, 1e6):
            for t in (1.0, 0.5, 1.0 / max(int(T), 1)):
                p = (torch.randn(B, K, generator=gen) * scale_p).requires_grad_(True)
                ref = _affine_probe(fn, p, t, ref)
                if ref is None:
                    return None
    except GaudiError:
        raise
    except Exception:
        return None
    if not bool((ref != 0).any()):
        return None
    return ref.numpy().astype(np.float32)


def affine_gradient_holds(fn, w, T: int) -> bool:
    """The closure that affine_target_weights recognised, held to its weight vector at EVERY t the chain passes to it --
    (s + 1) / T as float32, s = 0 .. T-1 (gaudi_hip.hip: sample_cb_impl) -- with rows of four magnitudes per call.  A closure
    whose python control flow depends on t (a guidance window) passes the three-point probe when the probe's points fall
    outside the window; this visits them all (T small torch evaluations; GaudiModel._run overlaps them with the device call)."""
    import torch
    K = int(len(w))
    ref = torch.from_numpy(np.asarray(w, np.float32))
    gen = torch.Generator().manual_seed(4321)
    base = torch.randn(4, K, generator=gen) * torch.tensor([[1.0], [1e-2], [30.0], [1e3]])
    try:
        for s in range(int(T)):
            t = float(np.float32(s + 1) / np.float32(T))
            if _affine_probe(fn, base.clone().requires_grad_(True), t, ref) is None:
                return False
    except GaudiError:
        raise
    except Exception:
        return False
    return True


def target_function_max_gap(cond_predictor) -> LinearTarget:
    """-pred[:,1]  (generation_guidance.py:200-203)."""
    w = np.zeros(cond_predictor.K, np.float32)
    w[1] = -1.0
    return LinearTarget(cond_predictor, w, name="max_gap")


def target_function_opv(cond_predictor, prop_dist: PropertyNorm) -> LinearTarget:
    """ip + ea + 3*gap on the un-normalised prediction (generation_guidance.py:205-211)."""
    w = np.zeros(cond_predictor.K, np.float32)
    w[0], w[2], w[3] = 3.0 * prop_dist.std[0], prop_dist.std[2], prop_dist.std[3]
    c = 3.0 * prop_dist.mean[0] + prop_dist.mean[2] + prop_dist.mean[3]
    return LinearTarget(cond_predictor, w, c, name="opv")


class GaudiModel:
    """Stands in for EnVariationalDiffusion on the sampling path: .sample / .sample_guidance / .normalize /
    .unnormalize / .T with the reference's signatures."""

    def __init__(self, args, state_dict, device: int = 0):
        self.args = checkpoint.args_dict(args)
        self.engine = Engine(device)
        self.engine.load_edm(self.args, state_dict)
        self.T = int(self.args["diffusion_steps"])
        self.in_node_nf = self.engine.F
        self.n_dims = 3
        self.norm_values = list(checkpoint.normalize_factors(self.args))
        self.norm_biases = (None, 0.0, 0.0)
        # Noise streams are keyed by (seed, global sample index, draw, element).  seed = None (default): every call draws its
        # 62-bit Philox key from torch's default generator, as the reference draws its noise from it (torch.randn,
        # en_diffusion.py:937-956): consecutive calls get fresh noise, torch.manual_seed(s) reproduces a run -- also when it
        # is called again with the SAME s between two calls (a stream position kept beside torch's seed
        # missed that) -- and anything else the caller draws from torch in between moves the key as it would move the
        # reference's noise.  An explicit seed (parity tests, sharded runs) keys the stream directly; then every call
        # consumes B fresh sample indices (sample_offset advances).
        self.seed = None
        self.sample_offset = 0
        self.injected_noise = None  # [T+2,B,N,3+F] raw draws (parity tests); None -> on-device Philox
        self.last_diag = None
        import weakref
        GaudiModel._latest = weakref.ref(self)

    _latest = None

    @classmethod
    def from_engine(cls, engine: Engine, args) -> "GaudiModel":
        """A model object over an Engine whose EDM weights are already loaded (bench.py: one handle serves several drivers)."""
        m = cls.__new__(cls)
        m.args = checkpoint.args_dict(args)
        m.engine = engine
        m.T = int(m.args["diffusion_steps"])
        m.in_node_nf = engine.F
        m.n_dims = 3
        m.norm_values = list(checkpoint.normalize_factors(m.args))
        m.norm_biases = (None, 0.0, 0.0)
        m.seed, m.sample_offset, m.injected_noise, m.last_diag = None, 0, None, None
        return m

    @classmethod
    def latest(cls):
        """The GaudiModel created last that is still alive (what get_cond_predictor_model(args, dataset) attaches to)."""
        m = cls._latest() if cls._latest is not None else None
        return m if m is not None and getattr(m.engine, "h", None) else None

    def _trace_closure(self, target):
        """An opaque target closure over (z, node_mask, edge_mask, t), as generation_guidance.py:198-211 writes them
        (``lambda z, nm, em, t: -cond_predictor(z, nm, em, t)[:, 1]``), turned into a PredTarget: while the closure runs, the
        predictor attached to this model returns a stand-in tensor, so torch.autograd yields dT/dpred for the GPU reverse
        pass.  The closure may be ANY differentiable torch function of the predictor outputs and t: affine ones run fused
        (LinearTarget), the others through the per-step callback (PredTarget); one that ALSO depends on z outside the
        predictor becomes a ZTarget (the host adds its direct dT/dz, gaudi_sample_cbz)."""
        cp = getattr(self, "cond_predictor", None)
        if cp is None:
            raise GaudiError("guidance with a target closure needs get_cond_predictor_model(...) on this model first")
        model = self

        def fn(pred, t):
            import torch
            B = pred.shape[0]
            # stand-in z (deterministic, non-zero: a direct dependence must show up in its gradient; torch's default generator
            # is not touched -- it keys the noise).  The stand-ins depend on (B, masks) only: built once per call, not per step.
            cache, key = model.__dict__.setdefault("_trace_cache", {}), B  # (_run empties it whenever it sets new masks)
            if cache.get("key") != key:
                shape = (B, model._trace_N, 3 + model.in_node_nf)
                nm_all, em_all = model._trace_nm, model._trace_em.reshape(model._trace_nm.shape[0], -1, 1)
                rows = np.arange(B) % nm_all.shape[0]  # (the affine probe runs on a few rows only)
                cache.update(key=key,
                             probe=(0.25 + torch.linspace(0.0, 1.0, int(np.prod(shape))).reshape(shape)).requires_grad_(True),
                             nm=torch.from_numpy(np.ascontiguousarray(nm_all[rows])),
                             em=torch.from_numpy(np.ascontiguousarray(em_all[rows]).reshape(-1, 1)))
            probe, nm, em = cache["probe"], cache["nm"], cache["em"]
            cp._override, cp._override_used = pred, False
            try:
                val = target(probe, nm, em, torch.full((B, 1), float(t)))
            finally:
                cp._override = None
            if not torch.is_tensor(val) or val.dim() == 0 or val.shape[0] != B:
                raise GaudiError("the target function must return one value per molecule (a torch tensor [B]); use a "
                                 "LinearTarget / PredTarget or a closure over cond_predictor (generation_guidance.py:198-211)")
            if val.requires_grad:
                (gz,) = torch.autograd.grad(val.sum(), probe, allow_unused=True, retain_graph=True)
                if gz is not None and bool((gz != 0).any()):
                    raise _DirectZ()  # depends on z outside cond_predictor too: the general form (ZTarget)
            # The closure must route through the predictor attached to THIS model: one that closes over another predictor (or a
            # torch module, or uses t only) would give dT/dpred = 0 here and the chain would run unguided without a word.
            if not cp._override_used or (pred.requires_grad and not val.requires_grad):
                raise GaudiError("the target function does not use the predictor attached to this model (it must call the "
                                 "cond_predictor returned by get_cond_predictor_model(..., model=<this model>)): its value does "
                                 "not depend on the prediction, guidance would be a no-op")
            return val

        pt = PredTarget(cp, fn, name=getattr(target, "__name__", "closure"))
        try:  # one probe evaluation: does the closure route through the predictor, does it depend on z directly?
            import torch
            fn(torch.zeros(min(2, int(self._trace_nm.shape[0])), cp.K).requires_grad_(True), 0.5)
        except _DirectZ:
            return ZTarget(cp, target, name=pt.name + " (direct z dependence)")
        lin = affine_target_weights(pt.fn, cp.K, self.T)
        if lin is not None:
            # affine in pred and independent of t (both closures the reference ships are, generation_guidance.py:200-211): the
            # declarative form, whose guidance is fused into the step kernel (one launch per 25 steps, no host round trip)
            lt = LinearTarget(cp, lin, name=pt.name + " (affine: fused)")
            lt._closure_fn = pt.fn  # _run re-checks the closure's gradient at the predictions the chain ends on
            return lt
        return pt

    def eval(self):
        return self

    # ---- en_diffusion.py:384-415
    def normalize(self, x, h, node_mask):
        x = _to_numpy(x) / self.norm_values[0]
        nm = _to_numpy(node_mask)
        h_cat = (_to_numpy(h["categorical"]).astype(np.float32) - self.norm_biases[1]) / self.norm_values[1] * nm
        return _like_ref(x.astype(np.float32)), {"categorical": _like_ref(h_cat.astype(np.float32)),
                                                  "integer": h.get("integer")}, None

    def unnormalize(self, x, h_cat, h_int, node_mask):
        nm = _to_numpy(node_mask)
        x = _to_numpy(x) * self.norm_values[0]
        h_cat = (_to_numpy(h_cat) * self.norm_values[1] + self.norm_biases[1]) * nm
        return _like_ref(x), _like_ref(h_cat), h_int

    def next_stream(self, n_samples: int):
        """-> (seed, sample_offset) for a call that draws noise for ``n_samples`` molecules; advances the offset."""
        if self.seed is None:
            import torch
            return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()), 0
        off = self.sample_offset
        self.sample_offset += int(n_samples)
        return int(self.seed), off

    def _run(self, n_samples, node_mask, edge_mask, std, target, scale, fix_noise):
        nm = _to_numpy(node_mask).astype(np.float32)
        B, N = nm.shape[0], nm.shape[1]
        if B != n_samples:
            raise GaudiError(f"n_samples={n_samples} but node_mask has batch {B}")
        em = _to_numpy(edge_mask).astype(np.float32).reshape(B, N, N)
        tw = None
        if target is not None:
            if not isinstance(target, (LinearTarget, PredTarget, ZTarget)):
                if not callable(target):
                    raise GaudiError("target_function must be callable")
                # the reference's own form: a closure over (z, node_mask, edge_mask, t) that calls cond_predictor
                self._trace_N, self._trace_nm, self._trace_em = N, nm.reshape(B, N, 1), em.reshape(B * N * N, 1)
                self._trace_cache = {}
                target = self._trace_closure(target)
            if target.cond_predictor.engine is not self.engine:
                raise GaudiError("the target's predictor must be attached to this model (get_cond_predictor_model(..., model=model))")
        seed, off = self.next_stream(B)
        # fix_noise (en_diffusion.py:562-566,972-978): one raw draw per call, broadcast over the batch
        self.engine.set_fix_noise(bool(fix_noise), off)
        try:
            if isinstance(target, ZTarget):
                nm3, em_flat = nm.reshape(B, N, 1), em.reshape(B * N * N, 1)
                x, h, diag = self.engine.sample_callback(nm.reshape(B, N), em, lambda z, p, t: target.grad(z, p, t, nm3, em_flat),
                                                         seed=seed, sample_offset=off, noise=self.injected_noise, std=std,
                                                         scale=scale, with_z=True)
            elif isinstance(target, PredTarget):
                x, h, diag = self.engine.sample_callback(nm.reshape(B, N), em, target.grad, seed=seed, sample_offset=off,
                                                         noise=self.injected_noise, std=std, scale=scale)
            else:
                tw = None if target is None else target.weights
                fn = getattr(target, "_closure_fn", None)
                holds, checker = [True], None
                if fn is not None:
                    # the closure was recognised as affine at three values of t; while the device runs the fused chain, the host
                    # holds it to that weight vector at every t of the chain (a closure with a guidance window).
                    # NOTE for callers: the closure is evaluated on a helper thread beside the engine call (it only ever sees
                    # stand-in predictions, never the engine) -- a closure with side effects on shared state must be wrapped in
                    # PredTarget.  The verdict is cached per (closure, T, weights): repeated calls of design() with the same
                    # closure pay for it once.
                    import threading
                    key = (id(fn), int(self.T), tw.tobytes())
                    cache = self.__dict__.setdefault("_affine_verdicts", {})
                    if key in cache and cache[key][0] is fn:
                        holds[0] = cache[key][1]
                    else:
                        def _check_all_t():
                            try:
                                holds[0] = affine_gradient_holds(fn, tw, self.T)
                            except BaseException:
                                holds[0] = False
                        checker = threading.Thread(target=_check_all_t, daemon=True)
                        checker.start()
                try:
                    out = self.engine.sample(nm.reshape(B, N), em, seed=seed, sample_offset=off, noise=self.injected_noise, std=std,
                                             target_w=tw, scale=scale, return_z0=fn is not None)
                finally:
                    if checker is not None:  # (joined whether or not the engine call raised)
                        checker.join()
                        if len(cache) > 64:
                            cache.clear()
                        cache[key] = (fn, holds[0])  # (the closure object is kept: its id cannot be reused while cached)
                x, h, diag = out[0], out[1], out[2]
                if fn is not None and not holds[0]:
                    # not the affine function the probe saw: the fused result is discarded and the SAME call (same noise stream)
                    # runs through the general path, which differentiates the closure at every step
                    pt = PredTarget(target.cond_predictor, fn, name=target.name + " -> callback")
                    x, h, diag = self.engine.sample_callback(nm.reshape(B, N), em, pt.grad, seed=seed, sample_offset=off,
                                                             noise=self.injected_noise, std=std, scale=scale)
                    diag = dict(diag, affine_recheck_failed=1)
                elif fn is not None:
                    # ... and at the predictions the chain actually ended on
                    import torch
                    t_last = 1.0 / self.T
                    p0 = torch.from_numpy(self.engine.predictor_fwd(out[3], t_last, nm.reshape(B, N), em)).requires_grad_(True)
                    with torch.enable_grad():
                        (g0,) = torch.autograd.grad(fn(p0, t_last).sum(), p0, allow_unused=True)
                    if g0 is None or not bool((g0 == torch.from_numpy(tw)[None]).all()):
                        raise GaudiError("the target closure was run as an affine function of the predictor outputs, but its "
                                         "gradient at the final predictions differs from the probed one: wrap it in "
                                         "PredTarget(cond_predictor, fn) to force the general path")
        finally:
            self.engine.set_fix_noise(False, 0)
        self.last_diag = diag
        F = h.shape[2]
        return _like_ref(x), {"categorical": _like_ref(h), "integer": _like_ref(np.zeros((B, N, 0), np.float32))}

    def sample(self, n_samples, n_nodes, node_mask, edge_mask, context=None, fix_noise=False, std=1.0):
        """EnVariationalDiffusion.sample (en_diffusion.py:958-1008)."""
        if context is not None:
            raise GaudiError("context conditioning is not part of the GaUDI sampling path")
        return self._run(n_samples, node_mask, edge_mask, std, None, 1.0, fix_noise)

    def sample_chain(self, n_samples, n_nodes, node_mask, edge_mask, context, keep_frames=None, std=1.0):
        """EnVariationalDiffusion.sample_chain (en_diffusion.py:1118-1174) -> chain_flat [keep_frames*n_samples, N, 3+F]."""
        if context is not None:
            raise GaudiError("context conditioning is not part of the GaUDI sampling path")
        nm = _to_numpy(node_mask).astype(np.float32)
        B, N = nm.shape[0], nm.shape[1]
        K = self.T if keep_frames is None else int(keep_frames)
        assert K <= self.T
        seed, off = self.next_stream(B)
        chain = self.engine.sample_chain(nm.reshape(B, N), _to_numpy(edge_mask).astype(np.float32).reshape(B, N, N), K,
                                         seed=seed, sample_offset=off, noise=self.injected_noise, std=std)
        return _like_ref(chain.reshape(K * B, N, chain.shape[-1]))

    def sample_guidance(self, n_samples, target_function, node_mask, edge_mask, scale=1, fix_noise=False, std=1.0):
        """EnVariationalDiffusion.sample_guidance (en_diffusion.py:1010-1067)."""
        return self._run(n_samples, node_mask, edge_mask, std, target_function, scale, fix_noise)


class CondPredictor:
    """Stands in for EGNN_predictor: callable (xh, node_mask, edge_mask, t) -> pred [B,K] (models.py:433-457)."""

    def __init__(self, model: GaudiModel, args, state_dict):
        self.args = checkpoint.args_dict(args)
        self.engine = model.engine
        self.engine.load_predictor(self.args, state_dict)
        self.K = self.engine.K

    @classmethod
    def from_engine(cls, model: GaudiModel, args) -> "CondPredictor":
        """Over a model whose engine already holds the predictor weights; attaches itself to the model."""
        cp = cls.__new__(cls)
        cp.args = checkpoint.args_dict(args)
        cp.engine = model.engine
        cp.K = model.engine.K
        model.cond_predictor = cp
        return cp

    def eval(self):
        return self

    _override = None  # set while a target closure is being traced (GaudiModel._trace_closure): the stand-in prediction

    def __call__(self, xh, node_mask, edge_mask, t=0.0):
        if self._override is not None:
            self._override_used = True
            return self._override
        z = _to_numpy(xh).astype(np.float32)
        B, N, _ = z.shape
        tt = np.broadcast_to(_to_numpy(t).astype(np.float32).reshape(-1), (B,)) if np.ndim(_to_numpy(t)) else float(t)
        return _like_ref(self.engine.predictor_fwd(z, tt, _to_numpy(node_mask).reshape(B, N),
                                                   _to_numpy(edge_mask).reshape(B, N, N)))


def get_model(args, dataloader_train=None, only_norm=True, device: int = 0, state_dict=None):
    """models_edm.get_model (models_edm.py:61-104) -> (model, nodes_dist, prop_dist).

    ``dataloader_train.dataset`` is only consulted for ``mean``/``std`` (property normalisation)."""
    a = checkpoint.args_dict(args)
    if state_dict is None:
        if not a.get("restore"):
            raise GaudiError("get_model needs a checkpoint (args.restore / exp_dir) or an explicit state_dict")
        state_dict = checkpoint.load_state_dict(a["exp_dir"])
    model = GaudiModel(a, state_dict, device=device)
    prop_dist = None
    ds = getattr(dataloader_train, "dataset", None)
    if ds is not None and getattr(ds, "mean", None) is not None:
        prop_dist = PropertyNorm(ds.mean, ds.std)
    return model, DistributionRings(a.get("dataset", "cata")), prop_dist


def get_cond_predictor_model(args, dataset=None, model: GaudiModel | None = None, state_dict=None) -> CondPredictor:
    """cond_prediction/train_cond_predictor.py:183-203 with the reference's own call shape ``(args, dataset)``: the predictor
    is attached to a GaudiModel's GPU handle so that guidance runs inside the same kernel launch as the denoiser --
    ``model`` if given, otherwise the model ``get_model`` created last (the order of generation_guidance.py:190-195)."""
    if model is None:
        model = GaudiModel.latest()
        if model is None:
            raise GaudiError("get_cond_predictor_model: call get_model first (the predictor shares the EDM's device handle) "
                             "or pass model=<GaudiModel>")
    a = checkpoint.args_dict(args)
    if state_dict is None:
        state_dict = checkpoint.load_state_dict(a["exp_dir"])
    model.cond_predictor = CondPredictor(model, a, state_dict)
    return model.cond_predictor
