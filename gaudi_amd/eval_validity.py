"""Host mirror of eval_validity.analyze_and_save (eval_validity.py:20-103) up to the RDKit / plotting calls:
sample ring counts from the dataset histogram, run the unguided sampler, compact by node_mask and check the
graph-of-rings stability of every molecule on the GPU."""
from __future__ import annotations

import math

from .analyze import analyze_validity_for_molecules
from .sampling_edm import sample_pos_edm


def analyze_and_save(args, model, nodes_dist, n_samples=1000, n_chains=0):
    """-> (stability_dict, molecule_list, molecule_stable_list).  RDKit validity / uniqueness, the plots and the
    visualisation chains (eval_validity.py:61-103) are not produced ("not planned")."""
    print("-" * 20)
    print("Generate molecules...")
    molecule_list = []
    n_samples = math.ceil(n_samples / args.batch_size) * args.batch_size
    for _ in range(n_samples // args.batch_size):
        nodesxsample = nodes_dist.sample(min(args.batch_size, n_samples))
        x, one_hot, node_mask, edge_mask = sample_pos_edm(args, model, nodesxsample)
        keep = [node_mask[i, :, 0].bool() for i in range(x.shape[0])]
        molecule_list += [(x[i][keep[i]], one_hot[i][keep[i]].argmax(dim=1)) for i in range(x.shape[0])]
    print(f"{len(molecule_list)} molecules generated, starting analysis")
    stability_dict, molecule_stable_list = analyze_validity_for_molecules(molecule_list, dataset=args.dataset,
                                                                          engine=model.engine)
    print(f"Stability for {getattr(args, 'exp_dir', '')}")
    for key, value in stability_dict.items():
        try:
            print(f"   {key}: {value:.2%}")
        except (TypeError, ValueError):
            pass
    return stability_dict, molecule_list, molecule_stable_list
