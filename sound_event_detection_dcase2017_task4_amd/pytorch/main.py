"""Train / inference CLI of the hot path — same sub-commands and flags as reference pytorch/main.py (:379-421), same
workspace layout (:72-110, :297-324) and checkpoint format (`{'iteration', 'model', 'optimizer'}`, :221-231).

    python -m sound_event_detection_dcase2017_task4_amd.pytorch.main train --dataset_dir D --workspace W \
        --holdout_fold 1 --model_type Cnn_9layers_FrameAvg --loss_type clip_bce --augmentation mixup \
        --learning_rate 1e-3 --batch_size 32 --resume_iteration 0 --stop_iteration 50000 --cuda

Differences, all additive: one process per GPU under torchrun (bucketed RCCL all-reduce of one flat gradient buffer)
instead of nn.DataParallel; `--synthetic N` trains on N synthetic clips when no packed data exists; `--print_every` (the
reference prints, i.e. synchronises, every iteration).

Batch semantics under N GPUs = the reference's (main.py:138, :160-166; nn.DataParallel scatters the batch):
`--batch_size` is the GLOBAL batch.  There is ONE seed-1234 sampler stream, ONE seed-1234 mixup-lambda stream and ONE
SpecAugment draw per step, computed identically on every rank; rank r consumes rows [r*n/N, (r+1)*n/N) of each (n =
waveforms per step; the slice length is even, so mixup pairs stay on one rank).  An N-rank run therefore sees exactly
the samples, lambdas and stripes of the 1-rank run with the same command line and differs from it only by per-replica
BatchNorm statistics -- which is also how DataParallel behaves.  `--per_gpu_batch` (extension) makes `--batch_size`
the per-GPU batch instead (global batch = N x batch_size; weak scaling).  The every-1000-iterations evaluation branch (main.py:188-209)
runs `evaluate.Evaluator` on the test / evaluation packs when they and their strong-label csv files exist (rank 0), with
the segment-based metrics restated in utils/utilities.py because sed_eval is not installed; otherwise it logs and skips.
"""
import argparse
import collections
import logging
import os
import pickle
import time

import numpy as np
import torch
import torch.utils.data

from .. import ops, parallel
from ..graph import GraphCaptureError, GraphedTrainStep
from ..optim import FusedAdamAmsgrad
from ..utils.config import (sample_rate, classes_num, mel_bins, fmin, fmax, window_size, hop_size)
from ..utils.augmentation import draw_specaug_stripes
from ..utils.data_generator import (DCASE2017Task4Dataset, PinnedBatchLoader, ShardedBatchSampler, TrainSampler, TestSampler,
                                    collate_fn)
from ..utils.utilities import (create_folder, get_filename, create_logging, Mixup, StatisticsContainer, random_state_to_plain,
                               random_state_from_plain)
from .evaluate import Evaluator
from . import models as _models
from .losses import get_loss_func
from .models import *  # noqa: F401,F403  (model lookup by name, like the reference's `eval(model_type)`)
from .pytorch_utils import move_data_to_device, do_mixup, forward


def _paths(args, prefix):
    sub = os.path.join('{}{}'.format(prefix, args.filename), 'holdout_fold={}'.format(args.holdout_fold),
                       'model_type={}'.format(args.model_type), 'loss_type={}'.format(args.loss_type),
                       'augmentation={}'.format(args.augmentation), 'batch_size={}'.format(args.batch_size))
    return (os.path.join(args.workspace, 'checkpoints', sub), os.path.join(args.workspace, 'logs', sub),
            os.path.join(args.workspace, 'predictions', sub))


def _build_model(model_type):
    assert model_type, 'Please specify model_type!'
    if model_type not in _models.__all__:
        raise Exception('Incorrect argument!')
    Model = getattr(_models, model_type)
    return Model(sample_rate, window_size, hop_size, mel_bins, fmin, fmax, classes_num)


POLL_LAG = 2          # optimiser steps between a step and the (rank-consistent) poll of its found-non-finite status
HIP_GRAPH_AUTO_MAX_CLIPS = 8       # --hip_graph auto: replay the step as one HIP graph when a rank trains on <= 8 clips per step
RECOVERIES = []       # one {'iteration', 'skipped', 'rank'} per recover() call of this process (what tests / wrappers inspect)


def train(args):
    rank, world, local_rank = parallel.init_from_env()
    device = 'cuda' if (args.cuda and torch.cuda.is_available()) else 'cpu'
    if device != 'cuda':
        raise SystemExit('This build runs the hot path on MI355X only: pass --cuda on a GPU box (no CPU fallback).')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    prefix = 'minidata_' if args.mini_data else ''
    checkpoints_dir, logs_dir, _ = _paths(args, prefix)
    if rank == 0:
        create_folder(checkpoints_dir)
        create_logging(logs_dir, 'w')
        logging.info(args)
        logging.info('Using GPU. ranks: {}'.format(world))
    loss_func = get_loss_func(args.loss_type)
    if args.synthetic:
        train_path = 'synthetic:{}'.format(args.synthetic)
    else:
        train_path = os.path.join(args.workspace, 'hdf5s', '{}training.h5'.format(prefix))
        if not os.path.exists(train_path) and os.path.isdir(train_path[:-3]):
            train_path = train_path[:-3]

    model = _build_model(args.model_type)
    iteration = 0
    ck = None
    if args.resume_iteration:
        # reference main.py:125-134 (which loads the model only and then fails on an undefined name): here the run CONTINUES --
        # model, optimiser moments and step, BatchNorm statistics, and the sampler / mixup / SpecAugment streams all pick up where
        # the checkpointed run was, so N iterations + resume + M iterations equal N + M iterations bit for bit (tests/test_gpu_cli.py)
        ck_path = os.path.join(checkpoints_dir, '{}_iterations.pth'.format(args.resume_iteration))
        if rank == 0:
            logging.info('Load resume model from {}'.format(ck_path))
        ck = torch.load(ck_path, map_location='cpu')
        model.load_state_dict(ck['model'])
        iteration = ck['iteration']
    model.to(device)
    # poll_lag: the found-non-finite guard is polled deterministically, POLL_LAG steps behind the newest one, so that every rank
    # learns about a refused step inside the same optimizer.step() call (optim.FusedAdamAmsgrad)
    optimizer = FusedAdamAmsgrad(model, lr=args.learning_rate, betas=(0.9, 0.999), eps=1e-08, world_size=world, poll_lag=POLL_LAG)
    if ck is not None and ck.get('optimizer'):
        # this build's flat moments, or the reference's torch.optim.Adam(amsgrad=True) state (main.py:222-230); anything else is
        # refused with the reason (optim.FusedAdamAmsgrad.load_state_dict) -- a resume never silently restarts the moments
        optimizer.load_state_dict(ck['optimizer'])
    streams = (ck or {}).get('streams') or {}
    if streams.get('torch_rng') is not None:
        torch.set_rng_state(streams['torch_rng'])          # SpecAugment draws continue (rank 0's state is broadcast below)
    parallel.broadcast_flat(optimizer.flat)
    parallel.broadcast_buffers(model)
    parallel.broadcast_rng_state()

    mix = 'mixup' in args.augmentation
    global_batch = args.batch_size * (world if args.per_gpu_batch else 1)          # post-mixup clips per step, all ranks
    rows_global = global_batch * 2 if mix else global_batch                        # waveforms per step, all ranks
    row_lo, row_hi = parallel.shard_rows(rows_global, rank, world, pair=mix)       # this rank's rows of every global batch
    # ONE sampler stream (seed 1234, data_generator.py:52-101) for the global batch; every rank walks it and loads only its
    # own rows.  Same batch stream as DataLoader(DCASE2017Task4Dataset, batch_sampler, collate_fn) (main.py:126-131 of the
    # reference), but assembled by threads into page-locked int16 buffers and uploaded one batch ahead on a copy stream: the
    # reference's 8 worker processes + pickled numpy batches deliver ~300-600 waveforms/s, one MI355X consumes 5300
    train_sampler = ShardedBatchSampler(TrainSampler(hdf5_path=train_path, batch_size=rows_global, random_seed=1234),
                                        row_lo, row_hi)
    # hold: a batch stays valid while POLL_LAG + 1 later ones are requested -- the batches of refused steps are re-run
    train_loader = PinnedBatchLoader(train_path, train_sampler, device=device, hold=POLL_LAG + 1)
    mixup_augmenter = Mixup(mixup_alpha=1., random_seed=1234) if mix else None
    if iteration:
        # the streams continue: the sampler has handed out `iteration` global batches, the mixup generator `iteration` lambda
        # vectors (its exact state is in the checkpoints this build writes; for a reference-written checkpoint it is replayed)
        train_sampler.sampler.skip(iteration)
        if mix and streams.get('mixup_rng') is not None:
            random_state_from_plain(mixup_augmenter.random_state, streams['mixup_rng'])
        elif mix:
            for _ in range(iteration):
                mixup_augmenter.get_lambda(batch_size=rows_global)
    # small per-GPU batches are launch-bound (--batch_size 32 over 8 GPUs = 4 clips each: ~140 kernels of 5-100 us): the step is
    # then replayed as ONE HIP graph unless --hip_graph off; larger batches keep the eager loop (no gain there)
    per_rank_clips = global_batch // world
    mode = getattr(args, 'hip_graph', 'auto')
    mode = {True: 'on', False: 'off', None: 'auto'}.get(mode, mode)       # programmatic callers: False has always meant off
    use_graph = mode == 'on' or (mode == 'auto' and per_rank_clips <= HIP_GRAPH_AUTO_MAX_CLIPS)
    graphed = GraphedTrainStep(model, optimizer, loss_func, mixup=mix) if use_graph else None
    if rank == 0:
        logging.info('HIP graph replay of forward + loss + backward: {} ({} clips per GPU)'.format('on' if use_graph else 'off', per_rank_clips))
        if use_graph and world > 1:
            logging.info('  (graph replay with %d ranks: the gradient buckets are all-reduced BEHIND the graph, not beside the '
                         'backward pass -- at <= %d clips per GPU the step is launch-bound and the exposed exchange is smaller '
                         'than the launch gaps it removes; --hip_graph off restores the overlapped exchange)',
                         world, HIP_GRAPH_AUTO_MAX_CLIPS)
    train_bgn_time = time.time()

    # evaluation sets of the every-1000-iterations branch (main.py:78-89, :150-176): used when present
    eval_sets = []
    if rank == 0 and not args.synthetic:
        _, _, predictions_dir = _paths(args, prefix)
        for data_type, pack, csv in (('test', '{}testing.h5'.format(prefix), 'groundtruth_strong_label_testing_set.csv'),
                                     ('evaluate', 'evaluation.h5', 'groundtruth_strong_label_evaluation_set.csv')):
            pack_path = os.path.join(args.workspace, 'hdf5s', pack)
            if not os.path.exists(pack_path) and os.path.isdir(pack_path[:-3]):
                pack_path = pack_path[:-3]
            csv_path = os.path.join(args.dataset_dir, 'metadata', csv)
            if os.path.exists(pack_path) and os.path.exists(csv_path):
                sampler = TestSampler(hdf5_path=pack_path, batch_size=args.batch_size)
                loader = PinnedBatchLoader(pack_path, sampler, device=device)
                eval_sets.append((data_type, loader, csv_path))
        if eval_sets:
            create_folder(predictions_dir)
            statistics_path = os.path.join(args.workspace, 'statistics', os.path.relpath(checkpoints_dir, os.path.join(
                args.workspace, 'checkpoints')), 'statistics.pickle')
            create_folder(os.path.dirname(statistics_path))
            statistics_container = StatisticsContainer(statistics_path)
            if args.resume_iteration and os.path.exists(statistics_path):
                statistics_container.load_state_dict(args.resume_iteration)        # keep the history up to the resume point
            evaluator = Evaluator(model=model)

    recent = collections.deque(maxlen=POLL_LAG + 1)      # (wave, target, lam, stripes) of the newest steps, oldest first

    graph_box = [graphed]

    def one_step(wave, target, lam, stripes):
        model.train()
        if graph_box[0] is not None:                 # same body, captured once per input shape and replayed
            try:
                return graph_box[0](wave, target, lam, stripes)
            except GraphCaptureError as err:
                # ONLY the refused capture lands here: nothing of this step has run yet, and with several ranks every rank is
                # here together (graph.GraphedTrainStep._capture_together).  Anything else -- an eager warm-up step, a replay,
                # optimizer.step() with its all-reduces and polls -- comes from a step that has been (partly) applied and
                # propagates: re-running it would update Adam and the BatchNorm statistics twice.
                if mode == 'on':
                    raise                            # asked for explicitly: fail loudly
                logging.warning('--hip_graph auto: the HIP graph capture failed (%r); continuing with the eager loop', err)
                graph_box[0] = None
                torch.cuda.synchronize()
        batch_output_dict = model(wave, lam, specaug_stripes=stripes)
        batch_target_dict = {'target': do_mixup(target, lam) if mix else target}
        step_loss = loss_func(batch_output_dict, batch_target_dict)
        optimizer.zero_grad()
        step_loss.backward()                         # gradient buckets are handed to RCCL as they complete
        optimizer.step()                             # waits for them; 1/world is folded into the Adam kernel
        return step_loss

    def recover(err, iteration):
        """ops.NonFiniteOperand out of optimizer.step() / optimizer.poll(): the Adam kernel refused the last
        `err.skipped_steps` optimiser steps -- on EVERY rank, and every rank is told inside the same call (deterministic lagged
        poll, rank flag on the last gradient bucket), so all of them arrive here together and the collectives stay balanced.
        Parameters, moments and BatchNorm running statistics are those from before the first refused step.  The reference has
        no guard (main.py:245-258: NaN flows into the weights); here the run continues on the fp32 MFMA kernels, which carry
        non-finite values exactly like the reference's torch ops, and the refused batches are run again on them, in order."""
        k = err.skipped_steps
        RECOVERIES.append({'iteration': iteration, 'skipped': k, 'rank': rank})
        logging.warning('iteration %d: %s', iteration, err)
        logging.warning('%d optimiser step(s) were refused on all %d rank(s); switching to the fp32 MFMA kernels '
                        '(ops.USE_SF16 = False) and re-running their batches', k, world)
        ops.rollback_bn_counters(model, k)
        ops.USE_SF16 = False
        redo = list(recent)[-k:] if k else []
        if len(redo) < k:
            raise RuntimeError('%d steps were refused but only %d batches are retained' % (k, len(redo)))
        loss = None
        for batch in redo:
            loss = one_step(*batch)
        return loss

    for batch_data_dict in train_loader:
        evaluate_now = iteration % 1000 == 0 and iteration > (args.resume_iteration or 0)
        checkpoint_now = iteration % (getattr(args, 'checkpoint_every', None) or 10000) == 0
        if evaluate_now or checkpoint_now:
            try:                                         # ALL ranks: nothing half-reported may reach an evaluation / checkpoint
                optimizer.poll(0)
            except ops.NonFiniteOperand as err:
                recover(err, iteration)
        if evaluate_now and rank == 0:
            train_fin_time = time.time()
            for data_type, loader, csv_path in eval_sets:
                statistics, _ = evaluator.evaluate(loader, csv_path, os.path.join(predictions_dir, '_tmp_submission.csv'))
                logging.info('{} statistics:'.format(data_type))
                logging.info('    Clipwise mAP: {:.3f}'.format(np.mean(statistics['clipwise_ap'])))
                if 'framewise_ap' in statistics:
                    logging.info('    Framewise mAP: {:.3f}'.format(np.mean(statistics['framewise_ap'])))
                logging.info('    {}'.format(statistics['sed_metrics']['overall']['error_rate']))
                statistics_container.append(data_type, iteration, statistics)
            if eval_sets:
                statistics_container.dump()
            logging.info('Iteration: {}  train time: {:.3f} s, validate time: {:.3f} s{}'.format(
                iteration, train_fin_time - train_bgn_time, time.time() - train_fin_time,
                '' if eval_sets else '  (no test / evaluation packs with strong-label csv found: evaluation skipped)'))
            train_bgn_time = time.time()
        if evaluate_now:
            parallel.barrier()           # the other ranks wait here (not inside an all-reduce) while rank 0 evaluates
        if checkpoint_now and rank == 0:
            # the reference's three keys (main.py:222-230) + 'streams' (additive; the reference reads 'model' and 'iteration'
            # only): what the NEXT iteration would draw from -- global torch generator (SpecAugment), mixup generator; the sampler
            # position is `iteration` batches into its seed-1234 stream
            checkpoint = {'iteration': iteration, 'model': model.state_dict(), 'optimizer': optimizer.state_dict(),
                          'streams': {'torch_rng': torch.get_rng_state(),
                                      'mixup_rng': random_state_to_plain(mixup_augmenter.random_state) if mix else None}}
            checkpoint_path = os.path.join(checkpoints_dir, '{}_iterations.pth'.format(iteration))
            torch.save(checkpoint, checkpoint_path)
            logging.info('Model saved to {}'.format(checkpoint_path))
        wave = batch_data_dict['waveform']               # int16 on the device (the log-mel kernel folds the /32767)
        target = batch_data_dict['target']
        if mix:                                          # global lambda stream, this rank's rows of it
            batch_data_dict['mixup_lambda'] = mixup_augmenter.get_lambda(batch_size=rows_global)[row_lo:row_hi]
        # SpecAugment positions of the GLOBAL batch from the global torch generator (all ranks hold the same state), rows
        # of this rank: the stripes do not depend on the number of ranks
        stripes = draw_specaug_stripes(rows_global, wave.shape[1] // hop_size + 1, mel_bins)[row_lo:row_hi]
        # (move_data_to_device's pageable copy would block the host until the GPU has drained: pinned staging instead)
        lam = ops.upload_small(batch_data_dict['mixup_lambda'], device, torch.float32) if mix else None
        recent.append((wave, target, lam, stripes))
        try:
            loss = one_step(wave, target, lam, stripes)
        except ops.NonFiniteOperand as err:
            loss = recover(err, iteration)
        if rank == 0 and args.print_every and iteration % args.print_every == 0:
            print(iteration, loss.item())
        if iteration == args.stop_iteration:
            break
        iteration += 1
    try:
        optimizer.poll(0)                                # steps still un-polled at the end (all ranks)
    except ops.NonFiniteOperand as err:
        recover(err, iteration)
    ops.check_device_errors(synchronize=True, nonfinite=True)
    parallel.shutdown()


def inference_prob(args):
    """Dump eval-mode probabilities of the test / evaluation packs to pickles (main.py:267-376, minus sed_eval)."""
    device = torch.device('cuda', 0) if (args.cuda and torch.cuda.is_available()) else None
    if device is None:
        raise SystemExit('This build runs the hot path on MI355X only: pass --cuda on a GPU box (no CPU fallback).')
    checkpoints_dir, _, predictions_dir = _paths(args, '')
    create_folder(predictions_dir)
    model = _build_model(args.model_type)
    checkpoint = torch.load(os.path.join(checkpoints_dir, '{}_iterations.pth'.format(args.iteration)), map_location='cpu')
    model.load_state_dict(checkpoint['model'])
    model.to(device)
    for data_type, name in (('test', 'testing.h5'), ('evaluate', 'evaluation.h5')):
        path = 'synthetic:{}'.format(args.synthetic) if args.synthetic else os.path.join(args.workspace, 'hdf5s', name)
        sampler = TestSampler(hdf5_path=path, batch_size=args.batch_size)
        loader = PinnedBatchLoader(path, sampler, device=device)      # int16 to the device; the log-mel kernel scales
        print('Inferencing {} data ...'.format(data_type))
        output_dict = forward(model, loader, return_target=True)
        prediction_path = os.path.join(predictions_dir, '{}_iterations.prediction.{}.pkl'.format(args.iteration, data_type))
        pickle.dump(output_dict, open(prediction_path, 'wb'))
        print('Write out to {}'.format(prediction_path))


def build_parser():
    parser = argparse.ArgumentParser(description='Example of parser. ')
    subparsers = parser.add_subparsers(dest='mode')
    p = subparsers.add_parser('train')
    p.add_argument('--dataset_dir', type=str, required=True, help='Directory of dataset.')
    p.add_argument('--workspace', type=str, required=True, help='Directory of your workspace.')
    p.add_argument('--holdout_fold', type=str, choices=['1'], required=True)
    p.add_argument('--model_type', type=str, required=True)
    p.add_argument('--loss_type', type=str, required=True)
    p.add_argument('--augmentation', type=str, choices=['none', 'mixup'], required=True)
    p.add_argument('--learning_rate', type=float, required=True)
    p.add_argument('--batch_size', type=int, required=True)
    p.add_argument('--resume_iteration', type=int)
    p.add_argument('--stop_iteration', type=int, required=True)
    p.add_argument('--cuda', action='store_true', default=False)
    p.add_argument('--mini_data', action='store_true', default=False)
    p.add_argument('--synthetic', type=int, default=0, help='(extension) train on N synthetic clips')
    p.add_argument('--print_every', type=int, default=100, help='(extension) loss print cadence; 1 = reference')
    p.add_argument('--hip_graph', nargs='?', const='on', default='auto', choices=['on', 'off', 'auto'],
                   help='(extension) replay forward + loss + backward as ONE HIP graph per step (graph.GraphedTrainStep): '
                        'frees the host from enqueueing ~140 kernels per step.  auto (default): on when a rank trains on '
                        '<= 8 clips per step (the launch-bound regime, e.g. --batch_size 32 over 8 GPUs), off otherwise')
    p.add_argument('--checkpoint_every', type=int, default=10000,
                   help='(extension, tests) checkpoint cadence in iterations; 10000 = reference (main.py:221)')
    p.add_argument('--per_gpu_batch', action='store_true', default=False,
                   help='(extension) --batch_size is per GPU (global batch = ranks x batch_size) instead of the global batch')
    q = subparsers.add_parser('inference_prob')
    q.add_argument('--dataset_dir', type=str, required=True, help='Directory of dataset.')
    q.add_argument('--workspace', type=str, required=True, help='Directory of your workspace.')
    q.add_argument('--holdout_fold', type=str, choices=['1'], required=True)
    q.add_argument('--model_type', type=str, required=True)
    q.add_argument('--loss_type', type=str, required=True)
    q.add_argument('--augmentation', type=str, choices=['none', 'mixup'], required=True)
    q.add_argument('--batch_size', type=int, required=True)
    q.add_argument('--iteration', type=int, required=True)
    q.add_argument('--cuda', action='store_true', default=False)
    q.add_argument('--synthetic', type=int, default=0)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    args.filename = get_filename(__file__)
    if args.mode == 'train':
        train(args)
    elif args.mode == 'inference_prob':
        inference_prob(args)
    else:
        raise Exception('Error argument!')


if __name__ == '__main__':
    main()
