"""`python -m kprn_amd.score <flags>` == `th eval/test_from_checkpoint.lua <flags>`
(release/songPathRnn/eval/model_test_one_list.sh:20).  Flags: test_from_checkpoint.lua:23-31 plus the
model-shape flags (the Torch7 checkpoint carried the module graph; the native checkpoint carries weights)."""
import argparse
import sys

from . import model, scoring


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-input_dir", default=""); p.add_argument("-out_file", default=""); p.add_argument("-predicate_name", default="")
    p.add_argument("-meanModel", type=int, default=0); p.add_argument("-model_path", default=""); p.add_argument("-test_list", default="")
    p.add_argument("-gpu_id", type=int, default=-1); p.add_argument("-top_k", type=int, default=2); p.add_argument("-k", type=int, default=5)
    args, rest = p.parse_known_args(argv)
    assert args.input_dir != "", "input_dir isnt set. Point to the dir where train/dev/test.list files reside"
    params = model.parse_flags(rest)
    params.topK, params.K, params.initModel, params.gpuid = args.top_k, args.k, args.model_path, args.gpu_id
    print("using model:", args.model_path)
    eng = model.build_engine(params)
    print({0: "Reducer is max pool", 1: "Reducer is topK", 2: "Reducer is log sum"}[args.top_k])
    print("start predicting...")
    scoring.test_from_checkpoint(eng, args.input_dir, args.test_list, args.out_file, log=sys.stdout)
    return 0


if __name__ == "__main__":
    sys.exit(main())
