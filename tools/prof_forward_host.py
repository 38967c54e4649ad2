"""cProfile of PredicateModel.forward() host work on the bench clip (inputs staged in HBM)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_amd.config import config_by_name
from laser_amd.predicate_model import PredicateModel
from laser_amd.synthetic import make_clip_batch
from laser_amd.tokenizer import HashTokenizer

cfg = config_by_name("vit-b-32")
tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
model = PredicateModel(None, None, "cuda:0", "vit-b-32", tokenizer=tok, init_seed=7).eval()
batch = make_clip_batch(100, n_videos=1, frames=32, objects=16, distinct_frames=4)
kw = batch.forward_kwargs()
staged = model.stage_inputs(batch.batched_videos, batch.batched_masks)
model(**kw, device_inputs=staged); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
out = model(**kw, device_inputs=staged)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
