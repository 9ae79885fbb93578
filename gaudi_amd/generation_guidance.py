"""Host mirror of generation_guidance.py: predict / get_target_function_values / design.  The reference filters the
designed molecules with RDKit validity (eval_stability, generation_guidance.py:69-80; dependency unavailable, SURVEY.md
section 8f "not planned"); here the filter is the graph-of-rings stability check of eval_validity.py on the GPU
(gaudi_amd.analyze), and plots are replaced by the returned dict."""
from __future__ import annotations

from time import time

import numpy as np

from .analyze import analyze_validity_for_molecules
from .models_edm import _like_ref, _to_numpy
from .sampling_edm import _check, sample_guidance


def _normalized_xh(x, h, node_mask, edm_model):
    x, hh, _ = edm_model.normalize(x, {"categorical": h, "integer": None}, node_mask)
    return np.concatenate([_to_numpy(x), _to_numpy(hh["categorical"])], axis=-1).astype(np.float32)


def predict(model, x, h, node_mask, edge_mask, edm_model):
    """generation_guidance.py:34-48: predictor at t=0 on the normalised sample."""
    bs, n_nodes, _ = _to_numpy(x).shape
    nm = _to_numpy(node_mask).reshape(bs, n_nodes, 1)
    em = _to_numpy(edge_mask).reshape(bs, n_nodes * n_nodes)
    return model(_normalized_xh(x, h, nm, edm_model), nm, em, np.zeros((bs, 1), np.float32))


def get_target_function_values(x, h, target_function, node_mask, edge_mask, edm_model):
    """generation_guidance.py:51-66."""
    bs, n_nodes, _ = _to_numpy(x).shape
    nm = _to_numpy(node_mask).reshape(bs, n_nodes, 1)
    em = _to_numpy(edge_mask).reshape(bs, n_nodes * n_nodes)
    # torch tensors, as the reference passes them: a target closure may combine them with the (torch) prediction
    return target_function(_like_ref(_normalized_xh(x, h, nm, edm_model)), _like_ref(nm.astype(np.float32)),
                           _like_ref(em.astype(np.float32)), _like_ref(np.zeros((bs, 1), np.float32)))


def eval_stability(x, one_hot, node_mask, edge_mask, dataset="cata", engine=None):
    """generation_guidance.py:69-80 with the graph-of-rings check in place of RDKit validity:
    -> (stability_dict, x, one_hot, node_mask, edge_mask) of the stable molecules."""
    import torch
    bs, n, _ = x.shape
    atom_type = one_hot.argmax(2)
    keep = [node_mask[i, :, 0].bool() for i in range(bs)]
    molecule_list = [(x[i][keep[i]], atom_type[i][keep[i]]) for i in range(bs)]
    stability_dict, _ = analyze_validity_for_molecules(molecule_list, dataset=dataset, engine=engine)
    ok = torch.tensor(stability_dict["molecule_stable_bool"], dtype=torch.bool)
    return stability_dict, x[ok], one_hot[ok], node_mask[ok], edge_mask.view(bs, n, n)[ok].view(-1, 1)


def design(args, model, cond_predictor, target_function, nodes_dist, prop_dist, scale, n_nodes):
    """generation_guidance.py:83-184: sample with guidance, check stability, evaluate the target and the predicted
    properties at t=0, rank all / stable molecules by target value.  Returns a dict instead of plotting."""
    model.eval()
    cond_predictor.eval()
    nodesxsample = np.array([n_nodes] * args.batch_size, dtype=np.int64)
    start_time = time()
    x, one_hot, node_mask, edge_mask = sample_guidance(args, model, target_function, nodesxsample, scale=scale)
    seconds = time() - start_time
    print(f"Generated {x.shape[0]} molecules in {seconds:.2f} seconds")
    _check(x, node_mask)
    stability_dict, _, _, _, _ = eval_stability(x, one_hot, node_mask, edge_mask, dataset=args.dataset,
                                                engine=model.engine)
    print(f"{scale=}")
    print(f"{stability_dict['mol_stable']=:.2%} out of {x.shape[0]}")
    tvals = _to_numpy(get_target_function_values(x, one_hot, target_function, node_mask, edge_mask, model))
    pred = _to_numpy(predict(cond_predictor, x, one_hot, node_mask, edge_mask, model))
    if prop_dist is not None:
        pred = prop_dist.unnormalize(pred)
    print(f"Mean target function value: {tvals.mean():.4f}")
    order = np.argsort(tvals)  # best (lowest energy) first, as the reference's ranking
    stable = np.array(stability_dict["molecule_stable_bool"], dtype=bool)
    if stable.any():
        print(f"Mean target function value (from stable): {tvals[stable].mean():.4f}")
    return dict(stability=stability_dict, best_stable=order[stable[order]], x=x, one_hot=one_hot, node_mask=node_mask, edge_mask=edge_mask, target_function_values=_like_ref(tvals),
                pred=_like_ref(pred), best=order, seconds=seconds, molecules_per_second=x.shape[0] / seconds)


def main(args, cond_predictor_args, prop_mean=None, prop_std=None, target="max_gap", batch_size=512, scale=0.6,
         n_nodes=10, device=0):
    """generation_guidance.main (reference lines 187-222) for checkpoint directories in the reference's format.

    ``args`` / ``cond_predictor_args`` come from ``gaudi_amd.checkpoint.get_edm_args / get_cond_predictor_args``
    (``args.txt`` + ``model.pt``).  The reference builds a dataloader only to obtain the property normalisation
    (``mean`` / ``std``, models_edm.py:111-112), which is not stored in checkpoints: pass it explicitly.  ``target`` is
    "max_gap" or "opv" (the two targets shipped with the reference) or a LinearTarget factory taking the predictor."""
    from .models_edm import (PropertyNorm, get_cond_predictor_model, get_model, target_function_max_gap,
                             target_function_opv)
    args.batch_size = batch_size  # the reference hard-codes these three knobs in main()
    model, nodes_dist, _ = get_model(args, device=device)
    cond_predictor = get_cond_predictor_model(cond_predictor_args, model=model)
    prop_dist = None
    if prop_mean is not None:
        prop_dist = PropertyNorm(prop_mean, prop_std)
    if target == "max_gap":
        target_function = target_function_max_gap(cond_predictor)
    elif target == "opv":
        if prop_dist is None:
            raise ValueError("the OPV target needs the property mean/std (prop_dist.unnormalize)")
        target_function = target_function_opv(cond_predictor, prop_dist)
    else:
        target_function = target(cond_predictor)
    return design(args, model, cond_predictor, target_function, nodes_dist, prop_dist, scale, n_nodes)
