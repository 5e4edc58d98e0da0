"""Interactive front end: ``python -m llm_consensus_amd.cli [--config FILE]``.

Same loop as the reference's ``main`` (/root/reference/src/main.rs:428-471): print
``Enter a question: ``, read one line, ``exit`` leaves, anything else is put to the
ensemble; afterwards the session state is reset and the prompt returns. Differences:

* the answer is written to stdout (the reference only logs it at info level,
  main.rs:464, so it is invisible without ``RUST_LOG=info``);
* there is no readiness poll: the reference spins on ``AnswerReadinessRequest`` every
  500 ms (main.rs:448-459) because its coordinator is an actor; here ``ask`` returns when
  the session has finished;
* end-of-file on stdin ends the loop (the reference would spin on empty questions);
* no API key is needed: the members run on local GPUs (main.rs:354-357 checks
  ``GEMINI_API_KEY``). ``CONSENSUS_CONFIG`` / ``--config`` select the ensemble and
  ``CONSENSUS_LOG`` (or ``RUST_LOG``) the log level.

``--questions FILE`` answers a file of questions (one per line) as one batch instead of
reading stdin, and ``--json`` prints one JSON object per answer with votes, rounds and
latency.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from typing import Iterable, Optional, TextIO

from .config import ConsensusConfig, load_config, load_config_or_exit  # noqa: F401  (load_config: re-exported)
from .utils.log import get_logger

PROMPT = "Enter a question: "
log = get_logger("consensus.cli")


def result_record(res) -> dict:
    return {
        "question": res.question,
        "answer": res.answer,
        "strategy": res.strategy,
        "rounds": res.rounds,
        "member_calls": res.member_calls,
        "unanimous": res.unanimous,
        "agreement": getattr(res, "agreement", None),
        "latency_s": round(res.latency_s, 6),
        "votes": getattr(res, "votes", None),
    }


def repl(ensemble, stdin: Optional[TextIO] = None, stdout: Optional[TextIO] = None, as_json: bool = False) -> int:
    """Runs the question loop until ``exit`` or end of input; returns the number of
    questions answered."""
    stdin, stdout = stdin or sys.stdin, stdout or sys.stdout
    answered = 0
    while True:
        stdout.write(PROMPT)
        stdout.flush()
        line = stdin.readline()
        if not line:
            stdout.write("\n")
            break
        question = line.strip()
        if question == "exit":
            break
        if not question:
            continue
        try:
            res = ensemble.ask(question)
        except Exception as e:  # a failed session must not take the loop down
            log.error("session failed: %s", e)
            stdout.write(f"System error: {e}\n")
            continue
        log.info("Final answer: %s", res.answer)
        stdout.write((json.dumps(result_record(res)) if as_json else res.answer) + "\n")
        stdout.flush()
        answered += 1
    return answered


def answer_file(ensemble, questions: Iterable[str], stdout: Optional[TextIO] = None, as_json: bool = False) -> int:
    stdout = stdout or sys.stdout
    qs = [q.strip() for q in questions if q.strip()]
    for res in ensemble.ask_many(qs):
        stdout.write((json.dumps(result_record(res)) if as_json else res.answer) + "\n")
    stdout.flush()
    return len(qs)


def main(argv: Optional[list] = None) -> int:
    ap = argparse.ArgumentParser(prog="llm-consensus", description=__doc__.split("\n\n")[0])
    ap.add_argument("--config", help="TOML/JSON/YAML ensemble description (default: $CONSENSUS_CONFIG, "
                                     "else four personas on one built-in model)")
    ap.add_argument("--questions", help="answer the questions of this file (one per line) as one batch")
    ap.add_argument("--json", action="store_true", help="print one JSON record per answer")
    ap.add_argument("--strategy", help="override the configured strategy")
    ap.add_argument("--device", help="cuda[:N] or cpu (default: GPU when present)")
    ap.add_argument("--no-warm-up", action="store_true",
                    help="skip the start-up warm-up (kernel loading, decode graph recording); the first "
                         "questions then pay for it")
    ap.add_argument("--gpus", type=int, default=None,
                    help="run as this many ranks, one per GPU, started by this command (default: the GPUs one "
                         "replica of the configured ensemble is planned for when several are visible; with "
                         "--device cpu the ranks are CPU processes)")
    ap.add_argument("--batched", action="store_true",
                    help="several ranks: answer each batch as one wave instead of serving continuously")
    ap.add_argument("--distributed", action="store_true",
                    help="this process is ONE rank of a group started by a launcher (this command with --gpus, "
                         "or torch.distributed.run): members are placed over the ranks by the planner; rank 0 "
                         "owns the terminal")
    argv = sys.argv[1:] if argv is None else list(argv)
    args = ap.parse_args(argv)

    cfg: ConsensusConfig = load_config_or_exit(args.config, "llm-consensus")
    if args.questions and not os.access(args.questions, os.R_OK):
        # (before any model is loaded: the file is read after the ensemble is up)
        print(f"llm-consensus: error: cannot read --questions {args.questions}", file=sys.stderr, flush=True)
        return 2
    random_init = sorted({m.model for m in cfg.members if not cfg.models[m.model].path})
    if random_init:
        # The reference refuses to start without a usable backend (main.rs:354-357). Random
        # weights are a usable engine but not a usable oracle: say so before the first answer.
        msg = (f"warning: model(s) {', '.join(random_init)} have no checkpoint path: running RANDOM-INIT weights, "
               "answers are noise (set models.<name>.path in --config / $CONSENSUS_CONFIG)")
        log.warning(msg)
        print(msg, file=sys.stderr, flush=True)
    if args.strategy:
        cfg.strategy = args.strategy
        try:
            cfg.validate()
        except ValueError as e:
            print(f"llm-consensus: error: --strategy: {e}", file=sys.stderr, flush=True)
            return 2
    from .parallel import launch

    if not args.distributed and not launch.under_launcher():
        # This process only supervises when the ensemble spans GPUs: it starts the ranks before it
        # has touched a GPU itself; rank 0 inherits the terminal.
        n = launch.ranks_for(cfg, args.gpus, args.device, replicate=bool(args.questions))
        if n > 1:
            return launch.launch_ranks("llm_consensus_amd.cli", launch.strip_flag(argv, "--gpus") + ["--distributed"], n)
        packed = launch.warn_if_packed(cfg, n, args.device)
        if packed:
            print(packed, file=sys.stderr, flush=True)
    elif launch.under_launcher():
        args.distributed = True
    import torch

    from .app import build_ensemble

    device = torch.device(args.device) if args.device else None
    leader = None
    if args.distributed:
        from .parallel.ensemble_parallel import init_distributed, serve_distributed

        comm, device = init_distributed(device.type if device else None)
        ensemble = leader = serve_distributed(cfg, comm, device, continuous=not args.batched)
        if ensemble is None:              # a follower rank: it has served until rank 0 closed
            return 0
    else:
        ensemble = build_ensemble(cfg, device)
    if not args.no_warm_up:
        ensemble.warm_up()
    if args.questions:
        with open(args.questions) as f:
            answer_file(ensemble, f, as_json=args.json)
    else:
        repl(ensemble, as_json=args.json)
    if leader is not None:
        leader.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
