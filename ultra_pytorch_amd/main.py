#!/usr/bin/env python3
"""Driver — counterpart of the reference's main.py (flags :24-62, train :85-227, test :230-292).

Same flags, same loop semantics: read train/valid data, derive `max_candidate_num` and `selection_bias_cutoff`,
pad, build the learning algorithm and the feeds from the class paths in the settings JSON (the plugin seam),
`get_batch(check_validation=True)` -> `model.train`, every `steps_per_checkpoint` steps run the valid set through
sequential `get_next_batch` / `model.validation`, merge batch-size weighted, save `state_dict()` as
`<model_dir>/<learning_algorithm>.ckpt` when `objective_metric` improves; the stop test is only evaluated at
checkpoint boundaries (Appendix A.12).  `--test_only` scores the test set and writes `<prefix>.ranklist`.

    python -m ultra_pytorch_amd.main --data_dir=./data/ --setting_file=settings.json --max_train_iteration=100
"""
import argparse
import copy
import json
import os
import sys
import time

import torch

from . import utils


def _str2bool(v):  # the reference uses type=bool (any non-empty string is True); accept real booleans too
    return str(v).lower() not in ("", "0", "false", "no")


def build_parser():
    p = argparse.ArgumentParser(description="Pipeline commandline argument")
    p.add_argument("--data_dir", type=str, default="./tests/data/")
    p.add_argument("--train_data_prefix", type=str, default="train")
    p.add_argument("--valid_data_prefix", type=str, default="valid")
    p.add_argument("--test_data_prefix", type=str, default="test")
    p.add_argument("--model_dir", type=str, default="./tests/tmp_model/")
    p.add_argument("--output_dir", type=str, default="./tests/tmp_output/")
    p.add_argument("--click_model_dir", type=str, default=None)
    p.add_argument("--data_format", type=str, default="ULTRA")
    p.add_argument("--setting_file", type=str, default="./example/offline_setting/dla_exp_settings.json")
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--max_list_cutoff", type=int, default=0)
    p.add_argument("--selection_bias_cutoff", type=int, default=10)
    p.add_argument("--max_train_iteration", type=int, default=10000)
    p.add_argument("--start_saving_iteration", type=int, default=0)
    p.add_argument("--steps_per_checkpoint", type=int, default=50)
    p.add_argument("--test_while_train", type=_str2bool, default=False)
    p.add_argument("--test_only", type=_str2bool, default=False)
    return p


def _ckpt_path(args, exp_settings):
    return os.path.join(args.model_dir, "%s.ckpt" % exp_settings["learning_algorithm"])


def create_model(args, exp_settings, data_set):
    model = utils.find_class(exp_settings["learning_algorithm"])(data_set, exp_settings)
    path = _ckpt_path(args, exp_settings)
    if os.path.isfile(path):
        print("Reading model parameters from %s" % path)
        model.model.load_state_dict(torch.load(path, map_location="cpu"))
        model.model.eval()
    else:
        print("Created model with fresh parameters.")
    return model


def _set_cutoffs(args, exp_settings):
    if "selection_bias_cutoff" not in exp_settings:
        exp_settings["selection_bias_cutoff"] = args.selection_bias_cutoff if args.selection_bias_cutoff > 0 else \
            exp_settings["max_candidate_num"]
    exp_settings["selection_bias_cutoff"] = min(exp_settings["selection_bias_cutoff"], exp_settings["max_candidate_num"])
    print("Users can only see the top %d documents for each query in training." % exp_settings["selection_bias_cutoff"])


def validate_model(model, data_set, feed, data_format):
    it, summaries, sizes = 0, [], []
    while it < len(data_set.initial_list):
        input_feed, info_map = feed.get_next_batch(it, data_set, check_validation=False, data_format=data_format)
        _, _, summary = model.validation(input_feed)
        summaries.append(copy.deepcopy(summary))
        sizes.append(len(info_map["input_list"]))
        it += sizes[-1]
    return utils.merge_Summary(summaries, sizes)


def train(args, exp_settings):
    print("Reading data in %s" % args.data_dir)
    rank_cut = args.max_list_cutoff if args.max_list_cutoff > 0 else None
    train_set = utils.read_data(args.data_dir, args.train_data_prefix, args.click_model_dir, rank_cut)
    valid_set = utils.read_data(args.data_dir, args.valid_data_prefix, args.click_model_dir, rank_cut)
    for ds in (train_set, valid_set):
        utils.find_class(exp_settings["train_input_feed"]).preprocess_data(ds, exp_settings["train_input_hparams"], exp_settings)
    print("Train Rank list size %d" % train_set.rank_list_size)
    print("Valid Rank list size %d" % valid_set.rank_list_size)
    exp_settings["max_candidate_num"] = max(train_set.rank_list_size, valid_set.rank_list_size)
    test_set = None
    if args.test_while_train:
        test_set = utils.read_data(args.data_dir, args.test_data_prefix, args.click_model_dir, rank_cut)
        utils.find_class(exp_settings["train_input_feed"]).preprocess_data(test_set, exp_settings["train_input_hparams"], exp_settings)
        exp_settings["max_candidate_num"] = max(test_set.rank_list_size, exp_settings["max_candidate_num"])
        test_set.pad(exp_settings["max_candidate_num"])
    _set_cutoffs(args, exp_settings)
    train_set.pad(exp_settings["max_candidate_num"])
    valid_set.pad(exp_settings["max_candidate_num"])

    print("Creating model...")
    os.makedirs(args.model_dir, exist_ok=True)
    model = create_model(args, exp_settings, train_set)
    train_feed = utils.find_class(exp_settings["train_input_feed"])(model, args.batch_size, exp_settings["train_input_hparams"])
    valid_feed = utils.find_class(exp_settings["valid_input_feed"])(model, args.batch_size, exp_settings["valid_input_hparams"])
    test_feed = None
    if args.test_while_train:
        test_feed = utils.find_class(exp_settings["test_input_feed"])(model, args.batch_size, exp_settings["test_input_hparams"])

    step_time, loss, current_step, best_perf = 0.0, 0.0, 0, None
    history = []
    print("max_train_iter: ", args.max_train_iteration)
    while True:
        start = time.time()
        input_feed, _ = train_feed.get_batch(train_set, check_validation=True, data_format=args.data_format)
        step_loss, _, _ = model.train(input_feed)
        step_time += (time.time() - start) / args.steps_per_checkpoint
        loss += step_loss / args.steps_per_checkpoint
        current_step += 1
        if current_step % args.steps_per_checkpoint != 0:
            continue
        print("global step %d learning rate %.4f step-time %.2f loss %.4f" % (model.global_step, model.learning_rate, step_time, loss))
        valid_summary = validate_model(model, valid_set, valid_feed, args.data_format)
        for key, value in valid_summary.items():
            print(key, value)
        if args.test_while_train:
            for key, value in validate_model(model, test_set, test_feed, args.data_format).items():
                print(key, value)
        history.append((current_step, loss, dict(valid_summary)))
        if "objective_metric" in exp_settings and exp_settings["objective_metric"] in valid_summary:
            value = valid_summary[exp_settings["objective_metric"]]
            if current_step >= args.start_saving_iteration and (best_perf is None or best_perf < value):
                torch.save(model.model.state_dict(), _ckpt_path(args, exp_settings))
                best_perf = value
                print("Save model, valid %s:%.3f" % (exp_settings["objective_metric"], best_perf))
        if best_perf is None and current_step > args.start_saving_iteration:
            torch.save(model.model.state_dict(), _ckpt_path(args, exp_settings))
        if loss == float("inf"):
            break
        step_time, loss = 0.0, 0.0
        sys.stdout.flush()
        if args.max_train_iteration > 0 and current_step > args.max_train_iteration:
            print("current_step: ", current_step)
            break
    return model, history


def test(args, exp_settings):
    print("Reading data in %s" % args.data_dir)
    rank_cut = args.max_list_cutoff if args.max_list_cutoff > 0 else None
    test_set = utils.read_data(args.data_dir, args.test_data_prefix, args.click_model_dir, rank_cut)
    utils.find_class(exp_settings["train_input_feed"]).preprocess_data(test_set, exp_settings["train_input_hparams"], exp_settings)
    exp_settings["max_candidate_num"] = test_set.rank_list_size
    _set_cutoffs(args, exp_settings)
    test_set.pad(exp_settings["max_candidate_num"])
    model = create_model(args, exp_settings, test_set)
    feed = utils.find_class(exp_settings["test_input_feed"])(model, args.batch_size, exp_settings["test_input_hparams"])
    it, rerank_scores, summaries, sizes = 0, [], [], []
    while it < len(test_set.initial_list):
        input_feed, info_map = feed.get_next_batch(it, test_set, check_validation=False)
        _, output_logits, summary = model.validation(input_feed)
        summaries.append(copy.deepcopy(summary))
        sizes.append(len(info_map["input_list"]))
        rows = output_logits.detach().cpu().tolist()
        rerank_scores.extend(rows[: sizes[-1]])
        it += sizes[-1]
    test_summary = utils.merge_Summary(summaries, sizes)
    print("  eval: %s" % " ".join("%s:%.3f" % kv for kv in test_summary.items()))
    os.makedirs(args.output_dir, exist_ok=True)
    utils.output_ranklist(test_set, rerank_scores, args.output_dir, args.test_data_prefix)
    return test_summary


def main(argv=None):
    args = build_parser().parse_args(argv)
    exp_settings = json.load(open(args.setting_file))
    return test(args, exp_settings) if args.test_only else train(args, exp_settings)


if __name__ == "__main__":
    main()
