"""Bare-name shim: `import layer_models as models` (QC/train_egcn.py:23; its model_dict, :85-94, names six classes)."""
from graph_odenet_amd.qc_models import (EdgeGCN_K_Set2Set, EdgeGCN_K_Sum, EdgeRES1_K_Set2Set,  # noqa: F401
                                        MPNN_ENN_K_Set2Set, MPNN_ENN_K_Sum, RESKnorm, UnimplementedModel,
                                        get_output_function)
