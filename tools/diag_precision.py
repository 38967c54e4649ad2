"""Diagnostic (not a test): feature / probability error of the HIP path against the
golden reference outputs, next to a torch emulation of bf16 rounding points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
from cases import FORWARD_CASES, build_case_batch
from golden_io import load_forward_case
from laser_amd.config import config_by_name
from laser_amd.predicate_model import PredicateModel
from laser_amd.tokenizer import HashTokenizer
from laser_amd.weights import predicate_model_state_dict

for name in ("tiny_multiclass_emb", "b32_infer", "tiny_infer"):
    cfg_name, wseed, _, fk = FORWARD_CASES[name]
    cfg = config_by_name(cfg_name)
    rec, cate, unary, binary, dummy, z = load_forward_case(name)
    tok = HashTokenizer(cfg.vocab, cfg.bos_token_id, cfg.eos_token_id)
    m = PredicateModel(None, None, "cuda:0", cfg_name, tokenizer=tok).eval()
    m.load_state_dict(predicate_model_state_dict(cfg, wseed, float(np.log(100.0))))
    fk2 = dict(fk); fk2["output_embeddings"] = True
    out = m(**build_case_batch(name).forward_kwargs(), **fk2)
    for v in cate:
        for nm, g, w in (("cate", out[0][v], cate[v]), ("unary", out[1][v], unary[v]), ("binary", out[2][v], binary[v])):
            if len(w) == 0: continue
            gv = np.array([float(g[k]) for k in w]); wv = np.array(list(w.values()))
            print(f"{name} vid{v} {nm}: max|dp|={np.abs(gv-wv).max():.4f} mean|dp|={np.abs(gv-wv).mean():.5f}")
    emb = out[4]
    for nm in ("cate_obj_clip_features", "unary_obj_clip_features", "binary_obj_pair_features"):
        key = f"{nm}_0"
        if key in z.files:
            got, want = emb[nm][0].float().cpu().numpy(), z[key]
            rel = np.linalg.norm(got - want, axis=-1) / np.linalg.norm(want, axis=-1)
            cos = (got * want).sum(-1) / (np.linalg.norm(got, axis=-1) * np.linalg.norm(want, axis=-1))
            print(f"{name} {nm}: rel err max {rel.max():.4e} mean {rel.mean():.4e}; 1-cos max {1-cos.min():.3e}; |f| {np.linalg.norm(want,axis=-1).mean():.3f}")
