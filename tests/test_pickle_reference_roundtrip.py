# -*- coding: utf-8 -*-
"""N2: model wire formats, checked against the REFERENCE where it is present.

The reference persists models with pickle; this package keeps its module paths / class names / attribute names,
so (a) the reference's pickles load here (golden G12, also tested on the GPU tier) and (b) pickles EMITTED here
load in the reference and score identically there.  (b) needs the reference itself: the test runs it in a
subprocess (its module names collide with the mirror package) and is skipped where /root/reference is absent
(the GPU box).  No GPU is needed: unpickling / pickling only moves parameters."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden

REF = "/root/reference"

_CHILD = r"""
import sys, types, pickle, importlib, io, contextlib, warnings
import numpy as np
sys.dont_write_bytecode = True
np.int = int
np.alltrue = np.all
pkg = types.ModuleType("sr")
pkg.__path__ = ["%s/sr"]
sys.modules["sr"] = pkg
R = importlib.import_module("sr.recognition")
blob, x = pickle.load(open(sys.argv[1], "rb"))
hmms = pickle.loads(blob)
assert type(hmms[0]).__module__ == "sr.recognition.hmm" and type(hmms[0]).__name__ == "HMM"
with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ev = [float(h.evaluate(x)) for h in hmms]
print(repr(ev))
""" % REF


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "sr", "recognition")), reason="reference not present")
def test_pickles_emitted_here_load_and_score_in_the_reference(tmp_path):
    g = load_golden("G12_reference_pickle")
    hmms = pickle.loads(g["pickle"].tobytes())           # reference pickle -> mirror classes
    import sr.recognition.hmm as mirror_hmm
    assert type(hmms[0]) is mirror_hmm.HMM
    blob = pickle.dumps(hmms, protocol=2)                 # mirror classes -> pickle, as continuous_train writes them
    job = tmp_path / "job.pkl"
    with open(job, "wb") as f:
        pickle.dump((blob, g["x"]), f)
    r = subprocess.run([sys.executable, "-B", "-c", _CHILD, str(job)], capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert r.returncode == 0, r.stderr
    ev = np.array(eval(r.stdout.strip().splitlines()[-1]))
    np.testing.assert_allclose(ev, g["evaluate"], rtol=1e-12)


def test_npz_model_format_round_trip(tmp_path):
    from sr.recognition.model_io import save_models_npz, load_models_npz, save_models_pickle, load_models_pickle
    g = load_golden("G12_reference_pickle")
    hmms = pickle.loads(g["pickle"].tobytes())
    p = str(tmp_path / "vocab.npz")
    save_models_npz(p, hmms)
    back = load_models_npz(p)
    assert len(back) == len(hmms)
    for a, b in zip(hmms, back):
        np.testing.assert_array_equal(a.transitions, b.transitions)
        for ga, gb in zip(a.gmm_states, b.gmm_states):
            np.testing.assert_array_equal(ga.w, gb.w)
            for da, db in zip(ga.dists, gb.dists):
                np.testing.assert_array_equal(da.mean, db.mean)
                np.testing.assert_array_equal(da.cov, db.cov)
    pp = str(tmp_path / "vocab.pkl")
    save_models_pickle(pp, back)
    again = load_models_pickle(pp)
    assert all(not (a != b) for x, y in zip(back, again) for a, b in zip(x.gmm_states, y.gmm_states))
