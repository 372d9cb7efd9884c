import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scream_amd.data import SyntheticPairs
from scream_amd.evaluate import evaluate_loader
from scream_amd.model import PointTransformer
from scream_amd.synthetic import make_state_dict
import multiprocessing as mp
def gen(i): return SyntheticPairs("3dmatch", 1, seed0=i)[0]
if __name__ == "__main__":
    with mp.get_context("spawn").Pool(14) as pool: items = pool.map(gen, range(96))
    class Mem(torch.utils.data.Dataset):
        def __len__(self): return len(items)
        def __getitem__(self, i): return items[i]
    net = PointTransformer(256, 6, 6); net.load_state_dict(make_state_dict(0, 256, 6, 6)); net = net.to("cuda:0").eval()
    evaluate_loader(net, Mem(), batch_pairs=32, verbose=False)
    pr = cProfile.Profile(); pr.enable()
    evaluate_loader(net, Mem(), batch_pairs=32, verbose=False)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
