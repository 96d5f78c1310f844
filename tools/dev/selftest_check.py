import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.set_device(0)           # HIP initialised BEFORE the package import: the environment is used as given
from graph_odenet_amd import hipgraph
print("env", os.environ.get(hipgraph.ENV), "-> memset_nodes_ok:", hipgraph.memset_nodes_ok())
