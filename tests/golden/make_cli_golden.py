"""Generates tests/golden/cli_reference.json from the REFERENCE's own io_utils.py / configs.py (importable on CPU here):
the default value of every command-line flag of train.py / test.py / train_regression.py / test_regression.py and the
checkpoint-file helpers' behaviour on a synthetic directory.  Re-run:  python tests/golden/make_cli_golden.py"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import configs as ref_configs  # noqa: E402
import io_utils as ref_io  # noqa: E402

out = {"flags": {}, "configs": {"kernel_type": ref_configs.kernel_type, "save_dir": ref_configs.save_dir}}
for script, fn in (("train", ref_io.parse_args), ("test", ref_io.parse_args), ("train_regression", ref_io.parse_args_regression),
                   ("test_regression", ref_io.parse_args_regression)):
    sys.argv = [script + ".py"]
    out["flags"][script] = vars(fn(script))
with tempfile.TemporaryDirectory() as d:
    res = {"empty_resume": ref_io.get_resume_file(d), "empty_best": ref_io.get_best_file(d)}
    for e in (0, 50, 7):
        open(os.path.join(d, "%d.tar" % e), "w").close()
    res["resume"] = os.path.basename(ref_io.get_resume_file(d))
    res["best_without_best_model"] = os.path.basename(ref_io.get_best_file(d))
    open(os.path.join(d, "best_model.tar"), "w").close()
    res["best_with_best_model"] = os.path.basename(ref_io.get_best_file(d))
    res["resume_with_best_model"] = os.path.basename(ref_io.get_resume_file(d))
    res["assigned_12"] = os.path.basename(ref_io.get_assigned_file(d, 12))
out["checkpoint_helpers"] = res
out["model_dict_keys"] = sorted(ref_io.model_dict.keys())
json.dump(out, open(os.path.join(HERE, "cli_reference.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
