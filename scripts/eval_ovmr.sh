#!/bin/bash
# Evaluation of a trained visual-token generator on the MI355X path with the command line of the reference's scripts/mm_cls/eval_ovmr.sh:
#
#   scripts/eval_ovmr.sh DATASET SEED SUB_CLASSES N_CTX EVAL_MODE EVAL_TAU GPU_ID
#
# The reference's two scripts build the same train.py command and differ in two directories (scripts/mm_cls/eval_ovmr.sh:25-29 against
# generate_classifier.sh:25-26): the generator's checkpoint comes from the base-to-new TRAINING output
#   output_ovmr/base2new/train_base/imagenet_21k_P/shots_64/MM_CLS_OP/<CFG>/seed1
# and the results (the evaluator's figures, per-class CSVs and, as a by-product of the first forward, mm_classifiers.pt / visual_tokens.pt)
# go to
#   output_ovmr/base2new/test_<SUB_CLASSES>_<EVAL_MODE>_tau<EVAL_TAU>/<DATASET>/shots_<SHOTS>/MM_CLS_OP/<CFG>/seed<SEED>
# (skipped when that directory exists).  Everything else -- environment, GPU selection, multi-rank launch, DRY_RUN=1 -- is
# scripts/generate_classifier.sh, which this script runs with those two directories; MODEL_DIR / DIR in the environment still win.
set -e
if [ $# -lt 7 ]; then
    echo "usage: $0 DATASET SEED SUB_CLASSES N_CTX EVAL_MODE EVAL_TAU GPU_ID" >&2
    exit 2
fi
DATASET=$1; SEED=$2; SUB=$3; EVAL_MODE=$5; EVAL_TAU=$6
TRAINER=MM_CLS_OP
CFG=${CFG:-vit_b16_c4_ep50_imagenet21k_pretrain}
SHOTS=${SHOTS:-16}
COMMON_DIR=${DATASET}/shots_${SHOTS}/${TRAINER}/${CFG}/seed${SEED}
COMMON_DIR_train=imagenet_21k_P/shots_64/${TRAINER}/${CFG}/seed1
export CFG SHOTS
export MODEL_DIR=${MODEL_DIR:-output_ovmr/base2new/train_base/${COMMON_DIR_train}}
export DIR=${DIR:-output_ovmr/base2new/test_${SUB}_${EVAL_MODE}_tau${EVAL_TAU}/${COMMON_DIR}}
exec bash "$(dirname "$0")/generate_classifier.sh" "$@"
