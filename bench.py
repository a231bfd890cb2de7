#!/usr/bin/env python3
"""bench.py -- decoded frames/sec of the batched beam-search decoder on MI355X.

Workload (BASELINE.json configs[1], "C2"): LexiconFreeDecoder + ZeroLM, CTC,
batch = 256 utterances per GPU, T = 1000, N = 29, beam = 50, beamSizeToken = 29,
beamThreshold = 25, logAdd = false, synthetic `ctc` emissions (SURVEY.md
Appendix A).  A "step" = one fltx_decode_batch over that batch: decodeBegin +
T frames + decodeEnd + back-trace of every hypothesis, inputs resident in HBM
before the timed region, results left in HBM.

--workload C3 / C4 / C5 select BASELINE.json configs[2] (90k-word trie, beam 50,
beamSizeToken 10), configs[3] (trie + synthetic 4-gram word LM, T = 1500, beam
100) and configs[4] (configs[3] with 8192 utterances over 8 GPUs = 1024 per GPU;
the line also carries the strong-scaling figure: 8192 utterances over the N GPUs
of the run).  The default and the judged line is C2.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
         --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; utterances shard across ranks with no data-path collective
(weak scaling: the same batch per GPU); torch.distributed is used only for the
barrier and the max-over-ranks of the timed region.  `--gpus N` WITHOUT the
launcher (WORLD_SIZE unset) starts the N ranks itself, so the line always
reports the N it was asked for.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : dominant kernel (the decode kernel) vs the HBM roofline: its own
                 algorithmic bytes (n-gram queries and
                 returned hypotheses as the kernels counted them) / its HIP-event
                 duration; the back-trace epilogue and the whole job likewise
  cpu_baseline : the unmodified reference (oracle/_ref, kind "reference") or, if
                 that prebuilt .so is absent, the oracle restatement (kind
                 "port"), one thread, on a bounded sample of the same batch;
                 N = 1 only, like the three blocks below
  cpu_baseline_steady / cpu_baseline_all_cores : the same decoder object reused
                 across utterances; one decoder per host thread on every core
  end_to_end   : host emissions in (H2D), kernels, the whole n-best back in host
                 memory, DecodeResult-like views materialised
  streaming    : the same batch as B parallel streams fed in chunks of 50 frames
                 with prune() after every chunk
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
HBM_MEASURED_GBS = 6290.0    # same guide: 6.29 TB/s float4 copy

WORKLOADS = {
    # name: (decoder, batch/GPU, T, beam, beamToken, LM)
    "C2": dict(lex=False, batch=256, T=1000, K=50, Kt=29, lm=False),
    "C3": dict(lex=True, batch=256, T=1000, K=50, Kt=10, lm=False),
    "C4": dict(lex=True, batch=256, T=1500, K=100, Kt=29, lm=True),
    "C5": dict(lex=True, batch=1024, T=1500, K=100, Kt=29, lm=True, total=8192),
    # word-piece token sets (the reference is N-agnostic: LexiconFreeDecoder.cpp:42-51 short-lists beamSizeToken
    # tokens per frame): lexicon-free + ZeroLM, beam 50, beamToken 50; N = 1024 by default, --tokens 8192 with --batch 64
    "WP": dict(lex=False, batch=256, T=1000, K=50, Kt=50, lm=False, tokens=1024),
    # C2's shape with a token-level n-gram LM (the mainstream use of LexiconFreeDecoder outside this benchmark:
    # LexiconFreeDecoder.cpp:69-85 + KenLM::score): a synthetic 3-gram over the 29 tokens, lmWeight 0.8
    "C2T": dict(lex=False, batch=256, T=1000, K=50, Kt=29, lm=False, toklm=(3, 11, 0.8)),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (0 = the workload's)")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--tokens", type=int, default=0, help="token-set size N (0 = the workload's: 29, WP: 1024)")
    ap.add_argument("--beam", type=int, default=0)
    ap.add_argument("--beam-token", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0, help="threads per utterance (0 = library default)")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2],
                    help="decoder objects (each on its own HIP stream) the steps alternate between: with 2 the "
                         "back-trace of one batch runs under the decode kernel of the next")
    ap.add_argument("--cpu-sample", type=int, default=160, help="utterances timed on the one-thread CPU baseline")
    ap.add_argument("--no-cpu", action="store_true", help="skip every CPU / host-side leg (profiling runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip all-cores / steady / end-to-end / streaming / sustained / secondary")
    ap.add_argument("--no-corun", action="store_true", help="with --pipeline 2: do not ask for the settings that let the decode "
                    "kernels of the two decoder objects share the CUs (defer_check, the 512-thread / shared-CU geometries)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 / C4 / C5-share / WP lines of the default run")
    ap.add_argument("--sustained-seconds", type=float, default=10.0, help="length of the back-to-back leg (0 = off)")
    ap.add_argument("--streaming-only", action="store_true", help="run the streaming leg alone and print its entry (profiling "
                    "a stream's kernels: rocprofv3 --kernel-trace --stats -- python bench.py --workload C2T --streaming-only)")
    ap.add_argument("--profile", action="store_true", help="print the per-phase clock split to stderr")
    ap.add_argument("--profile-out", default="", help="... and append it to this file (profiles/rNN/phase_split_*.txt)")
    ap.add_argument("--mode", default="process", choices=["process", "group"],
                    help="process: one rank per GPU (the driver's contract); group: ONE process, fltx_group_* over "
                         "--gpus devices (one context, decoder and host thread per device)")
    ap.add_argument("--profile-waves", default="0", help="comma-separated wave indices to sample with --profile")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--device", type=int, default=-1, help="override the device (default: LOCAL_RANK)")
    ap.add_argument("--set", action="append", default=[], help="decoder tunable key=value (repeatable)")
    ap.add_argument("--asg", action="store_true", help="ASG criterion (no blank, synthetic transitions; the lexicon without "
                                                        "doubled letters, as replabels guarantee) -- a secondary line")
    ap.add_argument("--log-add", action="store_true", help="logAdd merges -- a secondary line")
    return ap.parse_args()


def self_launch(a):
    """`--gpus N` without a launcher: start the N ranks (one per GPU) ourselves through
    torch.distributed.run and pass rank 0's line through, so n_gpus is what was asked for."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = [x for x in r.stdout.splitlines() if x.startswith("{")]
    if r.returncode != 0 or not line:
        sys.stdout.write(r.stdout)
        raise SystemExit("bench.py: the %d-rank run failed (exit %d)" % (a.gpus, r.returncode))
    print(line[-1])


TOL = [0.0]  # 1e-5 with --log-add (device libm: BASELINE.json's float-score tolerance), else bit equality


def nbest_same(got, want):
    if len(got) != len(want):
        return False
    for g, h in zip(got, want):
        if TOL[0] == 0.0:
            if not (g.score == h.score and g.am == h.am and g.lm == h.lm):
                return False
        elif abs(g.score - h.score) > TOL[0] or abs(g.am - h.am) > TOL[0] or abs(g.lm - h.lm) > TOL[0]:
            return False
        if not (np.array_equal(g.tokens, h.tokens) and np.array_equal(g.words, h.words)):
            return False
    return True


class Job:
    """Everything one rank needs to decode its shard: synthetic inputs, LM, trie, decoder."""

    def __init__(self, a, rank, local, B, cfg):
        from text_amd import _capi, synth
        self.capi = _capi
        self.a, self.B = a, B
        self.N = a.tokens or cfg.get("tokens", 29)
        self.T, self.K, self.Kt = cfg["T"], cfg["K"], min(cfg["Kt"], self.N)
        self.lex, self.haslm = cfg["lex"], cfg["lm"]
        N = self.N
        self.lexicon = synth.lexicon() if self.lex else None
        self.crit = "asg" if a.asg else "ctc"
        self.blank = -1 if a.asg else N - 1
        self.tr = synth.floats(4243, N * N, 0.0, 1.0) if a.asg else None
        if self.lex and a.asg:  # (a doubled letter makes "stay" and "advance" tie exactly: tests/helpers.py lexicon())
            sf, so = self.lexicon
            keep = [w for w in range(len(so) - 1) if not np.any(sf[so[w]:so[w + 1] - 1][1:] == sf[so[w]:so[w + 1] - 1][:-1])]
            nsf = np.concatenate([sf[so[w]:so[w + 1]] for w in keep]).astype(np.int32)
            nso = np.zeros(len(keep) + 1, dtype=np.int64)
            nso[1:] = np.cumsum([so[w + 1] - so[w] for w in keep])
            self.lexicon = (nsf, nso)
        self.dist = "lexspell" if self.lex else "ctc"
        self.u0 = rank * B
        self.e_host = synth.batch(self.dist, B, self.T, N, lexicon=self.lexicon, u0=self.u0)
        self.ctx = _capi.Context(device=local)
        self.ctx_device, self.more_ctx, self.more_tries = local, [], []
        self.lm = _capi.ZeroLM(self.ctx)
        self.opt = _capi.make_options(self.K, self.Kt, 25.0, 0.0, 0.0, float("-inf"), 0.0, bool(a.log_add), self.crit)
        self.arpa = None
        self.toklm = cfg.get("toklm")
        if self.toklm:
            order, seed, weight = self.toklm
            self.opt = _capi.make_options(self.K, self.Kt, 25.0, weight, 0.0, float("-inf"), 0.0, bool(a.log_add), self.crit)
            self.arpa = synthetic_token_arpa(N, order, seed)
            self.lm = _capi.ArpaLM(self.arpa[0], self.arpa[1])
        if self.haslm:
            # same options as the parity case C4_spell_u0 (tests/cases.py)
            self.opt = _capi.make_options(self.K, self.Kt, 25.0, 2.0, 2.0, float("-inf"), -1.0, bool(a.log_add), self.crit)
            self.arpa = synthetic_arpa(len(self.lexicon[1]) - 1)
            self.lm = _capi.ArpaLM(self.arpa[0], self.arpa[1])
        self.trie = self.host_trie = self.wscore = None
        self.W = 0
        if self.lex:
            W = self.W = len(self.lexicon[1]) - 1
            ht = self.host_trie = _capi.HostTrie(N, 0)
            self.wscore = np.zeros(W, dtype=np.float32)
            if self.arpa:  # trie label scores = lm.score(start, word) (DecoderTest.cpp:137-146)
                self.wscore = np.array([self.lm.score_sequence([w], False)[0][0] for w in range(W)],
                                       dtype=np.float32)
            ht.insert_many(self.lexicon[0], self.lexicon[1], np.arange(W), self.wscore)
            ht.smear(1)
            self.trie = ht.upload(self.ctx)
        self.Ts = np.full(B, self.T, dtype=np.int32)

    def close(self):
        """(a secondary workload's tables and contexts go before the next one is set up)"""
        for t in self.more_tries + ([self.trie] if self.trie is not None else []):
            t.close()
        self.lm.close()
        for c in self.more_ctx:
            c.close()
        self.ctx.close()

    def decoder(self, second_stream=False):
        """second_stream: a decoder on a context (= HIP stream) of its own, with its own copy of the
        trie; the LM object is shared (its tables are uploaded once per context)."""
        c = self.capi
        ctx, trie = self.ctx, self.trie
        if second_stream:
            ctx = c.Context(device=self.ctx_device)
            self.more_ctx.append(ctx)
            if self.lex:
                trie = self.host_trie.upload(ctx)
                self.more_tries.append(trie)
        if self.lex:
            d = c.BatchDecoder(ctx, c.LEXICON, self.opt, self.lm, 0, self.blank, unk=self.W, trie=trie, transitions=self.tr)
        else:
            d = c.BatchDecoder(ctx, c.LEXFREE, self.opt, self.lm, 0, self.blank, transitions=self.tr)
        if self.a.threads:
            d.set("threads", self.a.threads)
        if self.a.pipeline > 1 and not self.a.no_corun and not self.a.profile:
            # Two batches in flight, one per decoder object / HIP stream: a workgroup waits half of its cycles (the
            # frame step is a chain of LDS round trips and barriers), so a second utterance on the CU runs in those
            # gaps.  That needs (a) calls that return with their kernels queued -- the look at the utterances'
            # statuses is deferred to the first read of results ("defer_check"; the emissions stay in HBM), (b) the
            # geometries of which two workgroups fit a CU: 512 threads (8 waves x <= 128 VGPRs) instead of 576 for the
            # lexicon-free engine, the LM-state memo in HBM for the lexicon engines ("yshare"), and a back-trace
            # workgroup that leaves the LDS to them.
            d.set("defer_check", 1)
            d.set("bt_lds_kb", 64)
            if not self.lex and self.N <= 64 and self.K <= 64:
                d.set("slane_threads", 512)
            if self.lex:
                d.set("yshare", 1)
        for kv in self.a.set:
            k, v = kv.split("=")
            d.set(k, int(v))
        return d


def main():
    a = parse()
    TOL[0] = 1e-5 if a.log_add else 0.0
    if a.mode == "group":
        return group_mode(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decoder has no CPU path")
    if a.device >= 0:
        local = a.device
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants device %d but the node has %d (use --device 0 --backend gloo "
                         "only for plumbing checks: ranks sharing a GPU measure nothing)" %
                         (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)

    out, fail = measure(a, torch, dist, rank, local, world, primary=True)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if fail:
        raise SystemExit(fail)


def measure(a, torch, dist, rank, local, world, primary):
    """One workload through the timed loop -> (the JSON line's dict, None or why the line must not look green).
    primary: the judged line with every leg; otherwise a `secondary` entry (kernel timing, roofline, a small CPU
    sample for the n-best comparison)."""
    cfg = dict(WORKLOADS[a.workload])
    for key, val in (("T", a.frames), ("K", a.beam), ("Kt", a.beam_token)):
        if val:
            cfg[key] = val
    B = a.batch or cfg["batch"]
    job = Job(a, rank, local, B, cfg)
    T, N, K, Kt = job.T, job.N, job.K, job.Kt
    if a.streaming_only:
        return {"workload": a.workload, "streaming": streaming(job, B, T, N)}, None  # (no metric / value: not a bench line)
    e_dev = torch.from_numpy(job.e_host).cuda()  # resident in HBM before timing
    job.e_dev_ptr = e_dev.data_ptr()
    torch.cuda.synchronize()
    dec = job.decoder()
    decs = [dec] + [job.decoder(second_stream=True) for _ in range(a.pipeline - 1)]

    def fence():
        job.ctx.synchronize()
        for c in job.more_ctx:
            c.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def timed(step, steps, warmup, after=None):
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            if after:
                after()
        fence()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    kern_ms, bt_ms = [], []
    turn = [0]
    ran = [False] * len(decs)

    def read_events(i):  # HIP events the library recorded on that decoder's launch stream (waits for its batch)
        if ran[i]:
            d_ms, b_ms = decs[i].timing()
            kern_ms.append(d_ms)
            bt_ms.append(b_ms)
            ran[i] = False

    def step():  # one batch through the whole path; consecutive steps take turns on the decoder objects
        i = turn[0] % len(decs)
        read_events(i)  # (its previous batch: nothing the next launch would not wait for anyway)
        decs[i].decode_batch(None, job.Ts, N, device_ptr=e_dev.data_ptr())
        ran[i] = True
        turn[0] += 1

    # Set-up, not warm-up: every decoder object of the pipeline decodes the batch once (its first call sizes its
    # workspace -- tens of ms at big beams -- and with --warmup 1 the second object's would land in the timed region).
    for _ in range(len(decs)):
        step()
    fence()
    for i in range(len(decs)):
        read_events(i)
    del kern_ms[:], bt_ms[:]
    dt = timed(step, a.steps, a.warmup)
    for i in range(len(decs)):
        read_events(i)
    # (with "defer_check" = 1 every decode_batch above first settled the batch its decoder object had in flight: the
    # look at the utterances' statuses -- and the second pass of whatever a fast path flagged -- is inside the clock
    # for every timed batch; `unread_redone` counts what those looks decoded again)
    unread_redone = sum(d.get("unread_redone") for d in decs)
    in_region_k, in_region_b = list(kern_ms), list(bt_ms)
    kernels_only = None
    if primary and len(decs) > 1 and not a.no_corun and not a.profile and not a.no_extras:
        # the same loop with the looks dropped (defer_check = 2: kernels queued back to back, results left in HBM and
        # never looked at) -- round 5's headline, kept for comparison; it is not `value`
        for d in decs:
            d.get("redone")
            d.set("defer_check", 2)
        dt2 = timed(step, a.steps, a.warmup)
        for i in range(len(decs)):
            read_events(i)
        kernels_only = {"value": B * T * a.steps * world / dt2, "unit": "frames/s", "ms_per_step": dt2 / a.steps * 1e3,
                        "looks_dropped": sum(d.get("looks_dropped") for d in decs),
                        "note": "defer_check = 2: no status look, no second pass, results never read"}
        for d in decs:
            d.set("defer_check", 1)
        del kern_ms[:], bt_ms[:]
        kern_ms.extend(in_region_k)
        bt_ms.extend(in_region_b)
    # Per-kernel durations for the roofline: with two streams the events of the timed region also span the
    # time a kernel shares the CUs with the other stream's back-trace; a few launches on one stream, each
    # waited for, give the kernel's own duration (what rocprofv3's per-kernel average measures).
    over_k, over_b = list(kern_ms), list(bt_ms)
    if len(decs) > 1:
        del kern_ms[:], bt_ms[:]
        for _ in range(3):
            decs[0].decode_batch(None, job.Ts, N, device_ptr=e_dev.data_ptr())
            ran[0] = True
            read_events(0)
    if a.profile and rank == 0:
        phase_profile(a, dec, job, step, B, T)
    st = dec.stats()
    by = dec.bytes()
    engine = dec.get("engine")
    redone = max(d.get("redone") for d in decs)  # utterances of the last timed batches the engine handed to a more general one
    value = B * T * a.steps * world / dt

    out = {
        "metric": "decoded frames/sec (whole node), T=%d N=%d beam=%d; hyp bit-exact vs CPU" % (T, N, K),
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %s + %s CTC, batch=%d utterances/GPU, T=%d, N=%d, beam=%d, "
                               "beamToken=%d, beamThreshold=25, logAdd=false, `%s` emissions" %
                               (a.workload, "LexiconDecoder + 90k-word trie" if job.lex else "LexiconFreeDecoder",
                                "synthetic 4-gram word LM (lmWeight 2, wordScore 2, silScore -1)" if job.haslm
                                else ("synthetic token-level %d-gram LM (lmWeight %g)" % (job.toklm[0], job.toklm[2])
                                      if job.toklm else "ZeroLM"), B, T, N, K, Kt, job.dist),
                   "parallelism": "utterance-sharded x%d, no collective" % world,
                   "threads_per_utterance": st["threads_per_utt"], "lds_bytes_per_workgroup": st["lds_bytes"],
                   "engine": engine, "redone": redone, "unread_redone": unread_redone,
                   "status_look": "inside the timed region: every decode_batch settles the batch its decoder object had in "
                                  "flight (statuses read, flagged utterances decoded again) before it launches",
                   "lane_groups": dec.get("lane_groups"),
                   "why_not_lane": dec.get("why_not_lane"), "fallback_reasons": dec.get("fallback_reasons"),
                   "pipeline": "%d decoder object(s), one HIP stream each, taking turns batch by batch%s" % (
                       len(decs), " (the back-trace of a batch runs under the decode kernel of the next)"
                       if len(decs) > 1 else "")},
    }
    if kernels_only is not None:
        out["value_kernels_only"] = kernels_only
    # ---- roofline (rank-local): each kernel's own algorithmic bytes over its own duration ----
    k_ms, b_ms = float(np.mean(kern_ms)), float(np.mean(bt_ms))
    corun = len(decs) > 1 and not a.no_corun and not a.profile
    solo_ms = k_ms
    if corun:
        # the launches of the timed region overlap in pairs: a launch's own duration is the in-region average (what
        # rocprofv3's per-kernel average of this command shows); the chip's rate is the launches' bytes over the wall time
        k_ms = float(np.mean(over_k))
    ach = by["decode"] / (k_ms * 1e-3) / 1e9
    rl = {"bound": "hbm", "kernel": "fltx_decode_kernel", "achieved": ach, "peak": HBM_PEAK_GBS,
          "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
          "algorithmic_bytes_per_launch": by["decode"], "kernel_ms": k_ms,
          "frac_of_measured_copy_bw": ach / HBM_MEASURED_GBS, "us_per_frame_step": k_ms * 1e3 / T,
          "ngram_query_bytes": by["lm"],
          "kernel_ms_in_timed_region": float(np.mean(over_k)),
          "achieved_in_timed_region": by["decode"] / (float(np.mean(over_k)) * 1e-3) / 1e9,
          "frac_in_timed_region": by["decode"] / (float(np.mean(over_k)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
          "kernel_ms_alone": solo_ms, "concurrent_launches": 2 if corun else 1,
          "aggregate_achieved": by["decode"] * a.steps / dt / 1e9, "aggregate_frac": by["decode"] * a.steps / dt / 1e9 / HBM_PEAK_GBS,
          "timing": "HIP events on the launch stream; kernel_ms: %s" % (
              "the launches of the timed region, two in flight at a time on two streams (their workgroups share the CUs: "
              "kernel_ms_alone is one launch with the chip to itself, aggregate_* the launches' bytes over the wall time)"
              if corun else
              "3 launches on one stream right after the timed region (the timed region alternates two streams: "
              "kernel_ms_in_timed_region includes sharing the CUs with the other stream's back-trace)"
              if len(decs) > 1 else "the launches of the timed region")}
    e_ach = by["epilogue"] / (b_ms * 1e-3) / 1e9 if b_ms > 0 else 0.0
    rl["epilogue"] = {"kernel": "fltx_backtrace_kernel", "algorithmic_bytes_per_launch": by["epilogue"],
                      "kernel_ms": b_ms, "achieved": e_ach, "frac": e_ach / HBM_PEAK_GBS}
    w_ach = (by["decode"] + by["epilogue"]) / ((k_ms + b_ms) * 1e-3) / 1e9
    rl["whole_job"] = {"algorithmic_bytes_per_launch": by["decode"] + by["epilogue"], "kernels_ms": k_ms + b_ms,
                       "achieved": w_ach, "frac": w_ach / HBM_PEAK_GBS}
    # HBM traffic per launch comes from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    # over this same command (tools/pmc_traffic.py); it is quoted only for the geometry and
    # engine it was measured on.
    tr = pmc_traffic(a.workload, st["threads_per_utt"], B, T, N, K, engine)
    rl["frac_counter_bytes"] = None
    if tr is not None:
        rl["traffic"], rl["traffic_source"] = tr
        # what the hardware saw: HBM bytes of the PMC passes over the kernel's own duration, against the peak
        # (section 8(d)'s formula charges the lexicon decoder an edge gather the lane engines do not execute)
        rl["frac_counter_bytes"] = tr[0] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    out["roofline"] = rl

    # ---- configs[4]: the same 8192 utterances over however many GPUs run (strong scaling) ----
    total = cfg.get("total")
    if total and not a.batch:
        Bs = total // world
        job_s = job if Bs == B else Job(a, rank, local, Bs, cfg)
        es_dev = e_dev if Bs == B else torch.from_numpy(job_s.e_host).cuda()
        torch.cuda.synchronize()
        dec_s = dec if Bs == B else job_s.decoder()
        steps_s = max(1, a.steps // 4)
        dts = timed(lambda: dec_s.decode_batch(None, job_s.Ts, N, device_ptr=es_dev.data_ptr()), steps_s, 1)
        out["strong_scaling"] = {"total_utterances": Bs * world, "utterances_per_gpu": Bs, "steps": steps_s,
                                 "ms_per_step": dts / steps_s * 1e3, "value": Bs * world * T * steps_s / dts,
                                 "unit": "frames/s"}
        if dec_s is not dec:
            dec_s.close()

    # ---- sustained: back-to-back batches for >= --sustained-seconds (clock / thermal settling: the timed region
    # above is tens of milliseconds), frames/s over the whole leg and per window of 100 steps ----
    if primary and rank == 0 and world == 1 and a.sustained_seconds > 0 and not a.no_extras:
        win, rates = 100, []
        fence()
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < a.sustained_seconds or len(rates) < 2:
            t0 = time.perf_counter()
            for _ in range(win):
                step()
            fence()
            rates.append(B * T * win / (time.perf_counter() - t0))
        t_all = time.perf_counter() - t_start
        for i in range(len(decs)):
            read_events(i)
        out["sustained"] = {"seconds": t_all, "steps": win * len(rates), "value": B * T * win * len(rates) / t_all,
                            "unit": "frames/s", "window_steps": win, "window_min": min(rates), "window_max": max(rates),
                            "windows": len(rates), "first_window": rates[0], "last_window": rates[-1],
                            "note": "the timed loop's step, back to back; a fence (stream synchronize) after every window"}

    # ---- CPU baselines + parity spot check, end to end, streaming (rank 0, N=1 only) --------
    if rank == 0 and world == 1 and not a.no_cpu:
        dec.decode_batch(None, job.Ts, N, device_ptr=e_dev.data_ptr())  # the n-best compared below
        out["cpu_baseline"] = cpu_baseline(a, dec, job)
        if not primary and getattr(a, "secondary_legs", False):
            # (C3 / C4 as secondary workloads: their streaming and end-to-end legs under the driver's clock too)
            out["streaming"] = streaming(job, B, T, N)
            e2 = end_to_end(job, dec, B, T, N, more=False)
            out["end_to_end"] = e2
        if primary and not a.no_extras:
            out["cpu_baseline_steady"], out["cpu_baseline_all_cores"] = cpu_more(a, job)
            out["end_to_end"] = end_to_end(job, dec, B, T, N)
            out["end_to_end_two_streams"] = dict(out["end_to_end"]["two_streams"],
                                                 frac_of_value=out["end_to_end"]["two_streams"]["value"] / value)
            best_n = max(out["end_to_end"]["more_streams"], key=lambda k: out["end_to_end"]["more_streams"][k]["value"])
            out["end_to_end_best"] = dict(out["end_to_end"]["more_streams"][best_n], host_threads=int(best_n),
                                          frac_of_value=out["end_to_end"]["more_streams"][best_n]["value"] / value)
            out["streaming"] = streaming(job, B, T, N)
            if a.workload == "C2":
                try:
                    out["drop_in"] = drop_in(job, B, T, N)
                except Exception as ex:  # (the judged line does not depend on the binding module)
                    out["drop_in"] = {"error": repr(ex)}
    for d in decs:
        d.close()
    job.close()
    del e_dev
    # a line whose n-best differs from the reference's, or that spent its time in fallbacks,
    # must not look green
    fail = None
    mism = out.get("cpu_baseline", {}).get("gpu_nbest_mismatches_on_sample", 0) + \
        out.get("streaming", {}).get("final_nbest_mismatches_vs_cpu_on_sample", 0)
    if mism:
        fail = "bench.py: %s: %d of the sampled utterances differ from the CPU reference" % (a.workload, mism)
    if redone * 4 > B or unread_redone * 4 > B * (a.steps + a.warmup):
        fail = "bench.py: %s: %d of %d utterances fell back to a general engine (%d over the timed batches)" % (
            a.workload, redone, B, unread_redone)

    # ---- secondary: the other BASELINE configurations under the same clock (default invocation, N = 1) ----
    if primary and rank == 0 and world == 1 and a.workload == "C2" and not a.no_secondary and not a.no_extras and not a.no_cpu and \
            not (a.batch or a.frames or a.beam or a.beam_token or a.tokens or a.asg or a.log_add or a.set):
        import copy
        out["secondary"] = {}
        for name, wl, batch, sample in (("C3", "C3", 0, 24), ("C4", "C4", 0, 12), ("C5_share", "C5", 1024, 8),
                                        ("WP", "WP", 0, 8), ("C2_tokLM", "C2T", 0, 16), ("C2_tokLM_beam100", "C2T", 0, 8)):
            a2 = copy.copy(a)
            a2.workload, a2.batch, a2.cpu_sample = wl, batch, sample
            a2.beam = 100 if name.endswith("beam100") else a.beam  # (fltx_mlane.h's token-LM variant: two lane groups)
            a2.secondary_legs = name in ("C3", "C4", "C2_tokLM")  # (their streaming and end-to-end legs too)
            a2.steps, a2.warmup = max(12, a.steps), 2  # (two batches in flight: a short region is mostly ramp-up)
            t0 = time.perf_counter()
            try:
                o2, f2 = measure(a2, torch, dist, rank, local, world, primary=False)
            except Exception as e:  # noqa: BLE001 -- a secondary line must not take the judged line down with it
                out["secondary"][name] = {"error": repr(e)}
                fail = fail or "bench.py: secondary %s failed: %r" % (name, e)
                continue
            r2, c2 = o2["roofline"], o2.get("cpu_baseline", {})
            out["secondary"][name] = {
                "workload": o2["config"]["workload"], "value": o2["value"], "unit": "frames/s", "steps": o2["steps"],
                "ms_per_step": o2["ms_per_step"], "kernel_ms": r2["kernel_ms"], "engine": o2["config"]["engine"],
                "redone": o2["config"]["redone"], "why_not_lane": o2["config"]["why_not_lane"],
                "roofline": {"frac": r2["frac"], "frac_counter_bytes": r2["frac_counter_bytes"], "achieved": r2["achieved"],
                             "algorithmic_bytes_per_launch": r2["algorithmic_bytes_per_launch"], "traffic": r2["traffic"]},
                "mismatches_vs_reference_on_sample": c2.get("gpu_nbest_mismatches_on_sample"),
                "cpu_baseline_1thread": c2.get("value"), "cpu_sample_utterances": sample, "cpu_kind": c2.get("kind"),
                "streaming": {k: o2["streaming"].get(k) for k in ("value", "value_waiting_for_every_chunk", "engine",
                                                                  "stream_chunks_decoded_again",
                                                                  "final_nbest_mismatches_vs_cpu_on_sample")}
                if "streaming" in o2 else None,
                "end_to_end": {"one_stream": o2["end_to_end"]["value"], "two_streams": o2["end_to_end"]["two_streams"]["value"],
                               "unit": "frames/s"} if "end_to_end" in o2 else None,
                "wall_s_including_setup": time.perf_counter() - t0}
            fail = fail or f2
    return out, fail


def group_mode(a):
    """ONE process, the batch of --gpus x B utterances handed to fltx_group_decode_batch: the library cuts it into
    contiguous shards, one context / decoder / host thread per device, tables replicated, no inter-device traffic
    (SURVEY.md section 8e).  --device D puts every part on device D (plumbing check on a one-GPU box: not a scaling
    number).  Inputs resident in each device's HBM before the timed region."""
    import torch
    from text_amd import _capi
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decoder has no CPU path")
    cfg = dict(WORKLOADS[a.workload])
    for key, val in (("T", a.frames), ("K", a.beam), ("Kt", a.beam_token)):
        if val:
            cfg[key] = val
    B = a.batch or cfg["batch"]
    n = a.gpus
    devices = [a.device] * n if a.device >= 0 else list(range(n))
    if max(devices) >= torch.cuda.device_count():
        raise SystemExit("bench.py --mode group: device %d of %d" % (max(devices), torch.cuda.device_count()))
    from text_amd import synth
    j0 = Job(a, 0, devices[0], B, cfg)  # (host trie, LM, options; the group makes its own contexts and device tries)
    T, N, K, Kt = j0.T, j0.N, j0.K, j0.Kt
    e_hosts = [j0.e_host] + [synth.batch(j0.dist, B, T, N, lexicon=j0.lexicon, u0=i * B) for i in range(1, n)]
    bufs = []
    for i in range(n):
        with torch.cuda.device(devices[i]):
            bufs.append(torch.from_numpy(e_hosts[i]).cuda())
    torch.cuda.synchronize()
    grp = _capi.DecoderGroup(devices, _capi.LEXICON if j0.lex else _capi.LEXFREE, j0.opt, j0.lm, 0, j0.blank,
                             unk=j0.W if j0.lex else -1, host_trie=j0.host_trie, transitions=j0.tr)
    Ts = np.full(n * B, T, dtype=np.int32)
    offs = (np.arange(n * B, dtype=np.int64) % B) * T * N  # utterance b of part i sits at b - i * B in that device's buffer
    ptrs = [t.data_ptr() for t in bufs]

    def step():
        grp.decode_batch(None, Ts, N, offsets=offs, device_ptrs=ptrs)

    for _ in range(a.warmup):
        step()
    grp.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    grp.synchronize()
    dt = time.perf_counter() - t0
    parts = grp.parts()
    out = {"metric": "decoded frames/sec (whole node), T=%d N=%d beam=%d; hyp bit-exact vs CPU" % (T, N, K),
           "value": n * B * T * a.steps / dt, "unit": "frames/s", "n_gpus": n, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "%s, batch=%d utterances/GPU, T=%d, N=%d, beam=%d, beamToken=%d" % (a.workload, B, T, N, K, Kt),
                      "mode": "group: one process, fltx_group_decode_batch over devices %s (one context, decoder and host "
                              "thread per device, contiguous shards, no collective)" % devices,
                      "shards": [[f, c] for _, f, c in parts],
                      "same_device_plumbing_check": len(set(devices)) < n}}
    # parity spot check: first and last utterance of every shard against the reference CPU
    if not a.no_cpu:
        cpu = CpuSide(j0)
        mism = chk = 0
        for i, (_, first, count) in enumerate(parts):
            for b in (first, first + count - 1):
                d, lm = cpu.new_decoder()
                want = cpu.lib.decode(d, e_hosts[i][b - i * B], T, N)
                cpu.free(d, lm)
                got = grp.results(b)
                same = nbest_same(got, want)
                mism += 0 if same else 1
                chk += 1
        out["cpu_baseline"] = {"kind": cpu.kind, "gpu_nbest_mismatches_on_sample": mism, "utterances_checked": chk}
    print(json.dumps(out))
    grp.close()
    if out.get("cpu_baseline", {}).get("gpu_nbest_mismatches_on_sample", 0):
        raise SystemExit("bench.py --mode group: n-best differs from the CPU reference")


def phase_profile(a, dec, job, step, B, T):
    for pw in [int(x) for x in a.profile_waves.split(",")]:
        dec.set("profile", 1)
        dec.set("profile_wave", pw)
        dec.decode_batch(None, job.Ts, job.N, device_ptr=job.e_dev_ptr)  # (this decoder object: `step` takes turns)
        job.ctx.synchronize()
        pr = dec.profile().astype(np.float64)
        names = ["A2(combine)", "B(eval+bin)", "C(prefix)", "D(shortlist)", "E(build)", "row", "A1(relations)", "E-rank"]
        order = list(range(8))
        eng = dec.get("engine")
        if eng == 3:  # fltx_lane.h marks, in program order
            names = ["best+bins", "eval+hist", "bar1+prefix", "scatter", "build", "row+bar3", "load+relations", "bar2+rank"]
            order = [6, 0, 1, 2, 3, 7, 4, 5]
        if eng == 4:  # fltx_slane.h
            names = ["loads", "candidates+hist", "bar1+scan+select", "new-lane counts", "bar2+build", "bar3",
                     "(frames ranking the boundary bin)", "(members of that bin)"]
        if eng == 4 and dec.get("tlane"):  # ... its token-LM variant
            names = ["loads (LM gathers issued)", "bar0+bins+hist", "bar1+scan+select", "new-lane counts", "bar2+build", "bar3",
                     "re-entry", "candidates+LM wait+max"]
        if eng == 5:  # fltx_xlane.h (the last column is not clocks: per frame, boundary-bin rankings + 1e3 x
            # narrowed histogram passes + 1e6 x far-candidate counts)
            names = ["loads", "candidates+best", "barA+verdicts+hist", "bar1+scan+select", "new-lane counts",
                     "bar2+build", "bar3", "select paths"]
        if eng == 6:  # fltx_ylane.h
            names = ["loads", "candidates+best", "barA+verdicts+hist", "bar1+scan+select", "plans+counts", "bar2+build",
                     "bar3", "(word wave: up to the n-gram look-ups)"]
        names = [names[i] for i in order]
        pr = pr[order]
        tot = pr[:8].sum()
        line = "wave %d phase split (shader clocks, %% of %.3g): " % (pw, tot) + ", ".join(
            ("%s %.3f" if n == "select paths" or n.startswith("(fr") or n.startswith("(mem") or n.startswith("(word") else "%s %.0f") % (n, v / (B * T)) for n, v in zip(names, pr[:8])) + \
            " | clocks/frame/utt %.0f\n" % (tot / (B * T))
        sys.stderr.write(line)
        if a.profile_out:
            with open(a.profile_out, "a") as f:
                f.write("%s engine %d threads %d beam %d: %s" % (a.workload, eng, dec.get("threads"), job.K, line))
        dec.set("profile", 0)


def pmc_traffic(workload, threads, B, T, N, K, engine=None):
    import glob
    want = "%s threads=%d batch=%d T=%d N=%d beam=%d" % (workload, threads, B, T, N, K)
    if engine is not None:
        want += " engine=%d" % engine
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "r*", "hbm_traffic_*.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("workload") != want:
            continue
        for name, k in rec["kernels"].items():
            if "fltx_decode_kernel" in name:
                return k["hbm_bytes_per_launch"], os.path.relpath(path, here)
    return None


def synthetic_arpa(W):
    """Deterministic synthetic 4-gram ARPA file over the lexicon's words (SURVEY.md
    section 8d), cached under /tmp -> (path, vocabulary in LM-index order)."""
    from text_amd import ngram_synth
    vocab = ngram_synth.words(W) + ["<unk>"]
    d = os.environ.get("FLTX_NGRAM_CACHE", "/tmp/fltx_ngram_cache")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "lm_o4_s4242_v%d.arpa" % len(vocab))
    if not os.path.exists(path):
        tmp = "%s.tmp%d" % (path, os.getpid())
        ngram_synth.write_arpa(tmp, vocab[:-1], 4, (0, 200000, 100000, 50000), 4242)
        os.replace(tmp, path)
    return path, vocab


def synthetic_token_arpa(N, order, seed):
    """Deterministic synthetic n-gram over the N tokens (the file tests/helpers.py arpa_path builds for a
    ("ngram", order, seed) token LM) -> (path, vocabulary in token order)."""
    from text_amd import ngram_synth
    vocab = ngram_synth.words(N, "t")
    d = os.environ.get("FLTX_NGRAM_CACHE", "/tmp/fltx_ngram_cache")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "lm_o%d_s%d_v%d.arpa" % (order, seed, len(vocab)))
    if not os.path.exists(path):
        tmp = "%s.tmp%d" % (path, os.getpid())
        ngram_synth.write_arpa(tmp, vocab, order, (0, 3000, 1500, 800), seed)
        os.replace(tmp, path)
    return path, vocab


class CpuSide:
    """The CPU checker set up for the job's configuration (test infrastructure: only this
    baseline leg touches oracle/)."""

    def __init__(self, job):
        from oracle import orclib
        self.kind = "reference" if orclib.have_ref() else "port"
        self.lib = lib = orclib.load("ref" if self.kind == "reference" else "oracle")
        self.job = job
        self.opt = orclib.make_options(job.K, job.Kt, 25.0, 0.0, 0.0, float("-inf"), 0.0, bool(job.a.log_add), job.crit)
        if job.toklm:
            self.opt = orclib.make_options(job.K, job.Kt, 25.0, job.toklm[2], 0.0, float("-inf"), 0.0, bool(job.a.log_add), job.crit)
        elif job.arpa:
            self.opt = orclib.make_options(job.K, job.Kt, 25.0, 2.0, 2.0, float("-inf"), -1.0, bool(job.a.log_add), job.crit)
        self.trie = None
        if job.lexicon is not None:
            self.trie = lib.build_trie(job.N, 0, job.lexicon[0], job.lexicon[1], np.arange(job.W), job.wscore, 1)
        self.lm_shared = lib.lm_arpa_create(job.arpa[0].encode(), "\n".join(job.arpa[1]).encode()) \
            if job.arpa else None

    def new_decoder(self):
        lib, job = self.lib, self.job
        lm = self.lm_shared if self.lm_shared else lib.lm_zero_create()
        d = lib.lexicon(self.opt, self.trie, lm, 0, job.blank, job.W, job.tr, False) if self.trie \
            else lib.lexfree(self.opt, lm, 0, job.blank, job.tr)
        return d, lm

    def free(self, d, lm):
        self.lib.decoder_destroy(d)
        if not self.lm_shared:
            self.lib.lm_destroy(lm)


def cpu_baseline(a, dec, job):
    """Time the reference's CPU path on a bounded sample of the same batch and
    compare its n-best with what the GPU produced for those utterances."""
    cpu = CpuSide(job)
    n = min(a.cpu_sample, job.B)
    if job.arpa:
        n = min(n, 32)  # ~50 ms of CPU per frame-thousand at beam 100 with the 4-gram
    t_total = 0.0
    mism = 0
    for b in range(n):
        d, lm = cpu.new_decoder()
        t0 = time.perf_counter()
        hyps = cpu.lib.decode(d, job.e_host[b], job.T, job.N)  # fresh decoder per utterance, decode() only
        t_total += time.perf_counter() - t0
        cpu.free(d, lm)
        got = dec.results(b)
        same = nbest_same(got, hyps)
        mism += 0 if same else 1
    return {"value": n * job.T / t_total, "unit": "frames/s", "cores": 1, "kind": cpu.kind,
            "sample": "first %d utterances of the batch, one thread, fresh decoder per utterance, "
                      "decode() wall time only" % n,
            "seconds": t_total, "gpu_nbest_mismatches_on_sample": mism,
            "host_cpus": os.cpu_count()}


def cpu_more(a, job):
    """(steady state: one decoder object reused, as a server would; all cores: one decoder per
    host thread, trie / LM shared read-only, utterances dealt round-robin)."""
    from concurrent.futures import ThreadPoolExecutor
    cpu = CpuSide(job)
    lib = cpu.lib
    per_utt = 0.9 if job.arpa else (0.25 if job.lex else 0.09)  # rough seconds, to bound the legs
    # -- steady state, one thread
    n = max(4, min(job.B, int(8.0 / per_utt)))
    d, lm = cpu.new_decoder()
    lib.decode(d, job.e_host[0], job.T, job.N)
    t0 = time.perf_counter()
    for b in range(n):
        lib.decode(d, job.e_host[b % job.B], job.T, job.N)
    ts = time.perf_counter() - t0
    cpu.free(d, lm)
    steady = {"value": n * job.T / ts, "unit": "frames/s", "cores": 1, "kind": cpu.kind, "seconds": ts,
              "sample": "%d utterances through ONE decoder object (decodeBegin frees the previous "
                        "utterance's LM-state trie inside the timed region)" % n}
    # -- all cores (ctypes releases the GIL inside the call)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    quota = None
    try:  # a container may own fewer cores than it sees (cgroup v2 cpu.max = "quota period")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    visible = cores
    if quota:  # more threads than the quota only adds allocator contention
        cores = max(1, min(cores, int(quota + 0.999)))
    budget = 8.0  # seconds of wall time; every thread decodes whole utterances until it is spent
    done = [0] * cores
    t0 = time.perf_counter()

    def work(k):
        dd, ll = cpu.new_decoder()
        i = 0
        while time.perf_counter() - t0 < budget:
            lib.decode(dd, job.e_host[(k + i * cores) % job.B], job.T, job.N)
            i += 1
        done[k] = i
        cpu.free(dd, ll)

    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, range(cores)))
    ta = time.perf_counter() - t0
    allc = {"value": sum(done) * job.T / ta, "unit": "frames/s", "cores": cores, "kind": cpu.kind,
            "seconds": ta, "utterances": sum(done), "cgroup_cpu_quota_cores": quota, "visible_cpus": visible,
            "sample": "%d host threads, one decoder per thread (reused), trie / LM shared read-only, every "
                      "thread decodes utterances of the batch until %.0f s of wall time are spent" % (cores, budget)}
    return steady, allc


def drop_in(job, B, T, N, n_utt=48):
    """What a caller who only swaps the import gets: the reference's Python names (flashlight.lib.text.decoder through
    text_amd/compat), LexiconFreeDecoder.decode(ptr, T, N) utterance by utterance as bindings/python/test/test_decoder.py
    does -- host emissions in, a list of DecodeResult objects out, one launch per utterance on one CU -- and the same
    module's decode_batch over the whole batch."""
    compat = os.path.join(ROOT, "text_amd", "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    from flashlight.lib.text.decoder import CriterionType, LexiconFreeDecoder, LexiconFreeDecoderOptions, ZeroLM
    opts = LexiconFreeDecoderOptions(beam_size=job.K, beam_size_token=job.Kt, beam_threshold=25.0, lm_weight=0.0,
                                     sil_score=0.0, log_add=False, criterion_type=CriterionType.CTC)
    dec = LexiconFreeDecoder(opts, ZeroLM(), 0, N - 1, [])
    e = np.ascontiguousarray(job.e_host, dtype=np.float32).reshape(B, T * N)
    dec.decode(e[0].ctypes.data, T, N)  # (first call: context, buffers)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        nh = 0
        for b in range(n_utt):
            nh += len(dec.decode(e[b].ctypes.data, T, N))
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    out = {"api": "flashlight.lib.text.decoder.LexiconFreeDecoder.decode (pybind module of this repo, reference names)",
           "utterances": n_utt, "ms_per_utterance": best / n_utt * 1e3, "value": n_utt * T / best, "unit": "frames/s",
           "hypotheses_per_utterance": nh / n_utt,
           "note": "one utterance per call = one workgroup on one CU; the batch call of the same module is below"}
    # the reference's pattern for parallel decoding: one decoder object per thread (its own context = HIP stream here;
    # decode() releases the GIL while it waits for the device)
    import threading
    n_thr, per = 8, 12
    decs = [None] * n_thr
    count = [0] * n_thr

    def work(t, n):
        if decs[t] is None:
            decs[t] = LexiconFreeDecoder(opts, ZeroLM(), 0, N - 1, [])
        for i in range(n):
            count[t] += len(decs[t].decode(e[(t * per + i) % B].ctypes.data, T, N))

    for n in (1, per):  # (first pass: contexts, buffers)
        th = [threading.Thread(target=work, args=(t, n)) for t in range(n_thr)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
    out["one_decoder_per_thread"] = {"threads": n_thr, "utterances": n_thr * per, "ms_per_utterance": dt / (n_thr * per) * 1e3,
                                     "value": n_thr * per * T / dt, "unit": "frames/s"}
    Ts = [T] * B
    dec.decode_batch(e.ctypes.data, Ts, N)
    bb = None
    for _ in range(3):
        t0 = time.perf_counter()
        res = dec.decode_batch(e.ctypes.data, Ts, N)
        dt = time.perf_counter() - t0
        bb = dt if bb is None or dt < bb else bb
    out["decode_batch"] = {"utterances": B, "ms_per_batch": bb * 1e3, "value": B * T / bb, "unit": "frames/s",
                           "results": "list of lists of DecodeResult (%d objects)" % sum(len(r) for r in res)}
    t0 = time.perf_counter()
    view = dec.decode_batch_arrays(e.ctypes.data, Ts, N)
    dt = time.perf_counter() - t0
    out["decode_batch_arrays"] = {"utterances": len(view), "ms_per_batch": dt * 1e3, "value": B * T / dt, "unit": "frames/s",
                                  "results": "NumPy views, DecodeResult objects on demand"}
    # a user-defined LM (a Python subclass of LM, the reference's extension point: lm/LM.h:61-85, _decoder.cpp:39-56):
    # the search in the kernels, the LM's score() on the host once per frame for the frame's distinct questions --
    # BASELINE configs[0]'s shape (one utterance, T = 200, beam 10) with a ZeroLM written in Python
    try:
        from flashlight.lib.text.decoder import LM, LMState

        class PyZero(LM):
            def __init__(self):
                LM.__init__(self)
                self.calls = 0

            def start(self, start_with_nothing):
                return LMState()

            def score(self, state, idx):
                self.calls += 1
                return state.child(idx), 0.0

            def finish(self, state):
                return state, 0.0

        T1, K1 = min(T, 200), 10
        o1 = LexiconFreeDecoderOptions(beam_size=K1, beam_size_token=job.Kt, beam_threshold=25.0, lm_weight=0.0,
                                       sil_score=0.0, log_add=False, criterion_type=CriterionType.CTC)
        lm = PyZero()
        du = LexiconFreeDecoder(o1, lm, 0, N - 1, [])
        dz = LexiconFreeDecoder(o1, ZeroLM(), 0, N - 1, [])
        e1 = np.ascontiguousarray(job.e_host[0, :T1], dtype=np.float32)
        ru = du.decode(e1.ctypes.data, T1, N)
        rz = dz.decode(e1.ctypes.data, T1, N)
        same = len(ru) == len(rz) and all(a.score == b.score and list(a.tokens) == list(b.tokens) for a, b in zip(ru, rz))
        lm.calls = 0
        t0 = time.perf_counter()
        for _ in range(3):
            du.decode(e1.ctypes.data, T1, N)
        dtu = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3):
            dz.decode(e1.ctypes.data, T1, N)
        dtz = (time.perf_counter() - t0) / 3
        out["user_lm"] = {"lm": "a ZeroLM written in Python (subclass of flashlight.lib.text.decoder.LM)", "frames": T1, "beam": K1,
                          "ms_per_utterance": dtu * 1e3, "us_per_frame": dtu / T1 * 1e6, "lm_score_calls_per_utterance": lm.calls // 3,
                          "same_nbest_as_the_device_zero_lm": bool(same), "device_zero_lm_ms_per_utterance": dtz * 1e3,
                          "note": "two launches + one stream synchronize + the Python calls per frame"}
    except Exception as ex:  # noqa: BLE001 -- (the judged line does not depend on this leg)
        out["user_lm"] = {"error": repr(ex)}
    return out


def end_to_end(job, dec, B, T, N, more=True):
    """Host buffers on both sides: pageable emissions in (H2D inside the call), kernels, the n-best back in
    host memory through pinned staging (compacted on the device: the rows that exist, tokens as bytes), NumPy
    views over it (what a Python caller gets), and -- separately -- one Python object per hypothesis.  Then the
    same with two decoder objects driven by two host threads: the copies of one batch run under the kernels of
    the other."""
    import threading
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        dec.decode_batch(job.e_host, job.Ts, N)
        t1 = time.perf_counter()
        arrays = dec.results_arrays_compact()
        t2 = time.perf_counter()
        objs = dec.results_batch()
        t3 = time.perf_counter()
        cur = (t2 - t0, t1 - t0, t2 - t1, t3 - t2, len(objs), int(arrays["n_hyp"].sum()))
        best = cur if best is None or cur[0] < best[0] else best
    out = {"ms_per_batch": best[0] * 1e3, "value": B * T / best[0], "unit": "frames/s",
           "h2d_plus_kernels_ms": best[1] * 1e3, "results_to_host_arrays_ms": best[2] * 1e3,
           "python_object_per_hypothesis_ms": best[3] * 1e3, "hypotheses": best[5],
           "note": "value = emissions in host memory -> n-best in host memory as NumPy arrays (scores [B,K,3]; "
                   "the rows of the hypotheses that exist, tokens as uint8, packed on the device: "
                   "fltx_result_fetch_batch_compact); per-hypothesis Python objects are extra and optional"}
    # two batches in flight: one decoder object (own stream, own buffers) per host thread
    ds = [dec, job.decoder(second_stream=True)]
    n_each = 6

    def run(d):
        for _ in range(n_each):
            d.decode_batch(job.e_host, job.Ts, N)
            d.results_arrays_compact()

    run(ds[1])
    th = [threading.Thread(target=run, args=(d,)) for d in ds]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = (time.perf_counter() - t0) / (2 * n_each)
    ds[1].close()
    out["two_streams"] = {"ms_per_batch": dt * 1e3, "value": B * T / dt, "unit": "frames/s",
                          "note": "two decoder objects / HIP streams / host threads taking batches in turn: "
                                  "H2D and D2H of one batch under the kernels of the other"}
    # ... and with three and four: a batch's H2D (0.6 ms), kernels (2.1 + 0.3 ms with two launches sharing the CUs) and
    # D2H (0.45 ms) are a 3.4 ms chain per host thread, so two threads cannot keep two launches in flight all the time
    out["more_streams"] = {}
    for n_thr in ((3, 4) if more else ()):
        ds = [dec] + [job.decoder(second_stream=True) for _ in range(n_thr - 1)]
        for d in ds[1:]:
            run(d)
        th = [threading.Thread(target=run, args=(d,)) for d in ds]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dtn = (time.perf_counter() - t0) / (n_thr * n_each)
        for d in ds[1:]:
            d.close()
        out["more_streams"][str(n_thr)] = {"ms_per_batch": dtn * 1e3, "value": B * T / dtn, "unit": "frames/s"}
    return out


def streaming(job, B, T, N, chunk=50):
    """Decoder::decodeStep in chunks with prune() after each: B parallel
    streams, emissions from host memory chunk by chunk, getBestHypothesis at the end."""
    c = job.capi
    d = job.decoder()
    Tc = np.full(B, chunk, dtype=np.int32)
    nchunk = T // chunk
    lat = []
    best = best_wait = None
    # the chunks as a caller's audio front end would hand them over: contiguous [B, chunk, N] host arrays
    pieces = [np.ascontiguousarray(job.e_host[:, k * chunk:(k + 1) * chunk, :]) for k in range(nchunk)]
    d.set("stream_total_frames", T)  # (LM-state ids are created for as long as the stream runs: T frames, not the buffer's)
    for rep in range(3):
        # rep 0, 1: every chunk waited for (its latency); rep 2: the caller hands over the next chunk as soon as the
        # call returns -- the chunk's H2D then runs under the previous chunk's kernel (copy stream, two slots)
        wait_each = rep < 2
        d.stream_begin(B, N, 4 * chunk + 8)
        job.ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(nchunk):
            tc = time.perf_counter()
            d.stream_step(pieces[k], Tc)
            d.stream_prune(0)
            if wait_each:
                job.ctx.synchronize()
                lat.append(time.perf_counter() - tc)
        d.stream_end()
        job.ctx.synchronize()
        dt = time.perf_counter() - t0
        if wait_each:
            best_wait = dt if rep == 0 else min(best_wait, dt)
        else:
            best = dt
    redone = d.get("stream_redone")
    # the streams' final n-best (what is left in the buffer after the last prune) against the reference CPU fed the
    # same chunks, on the first streams; fetching a result also raises if a stream's status is not clean
    cpu = CpuSide(job)
    mism, n_chk = 0, min(B, 2 if job.arpa else 4)
    for b in range(n_chk):
        rd, rlm = cpu.new_decoder()
        cpu.lib.decoder_begin(rd)
        for k in range(nchunk):
            cpu.lib.decoder_step(rd, pieces[k][b].ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_float)),
                                 chunk, N)
            cpu.lib.decoder_prune(rd, 0)
        cpu.lib.decoder_end(rd)
        want = cpu.lib.collect(rd)
        cpu.free(rd, rlm)
        got = d.results(b)
        same = nbest_same(got, want)
        mism += 0 if same else 1
    for b in range(n_chk, B, max(1, B // 16)):
        d.results(b)  # (status check)
    d.close()
    lat = np.sort(np.array(lat[len(lat) // 2:]))
    return {"value": B * nchunk * chunk / best, "unit": "frames/s", "streams": B, "chunk_frames": chunk,
            "value_waiting_for_every_chunk": B * nchunk * chunk / best_wait,
            "chunk_latency_ms_median": float(np.median(lat) * 1e3),
            "chunk_latency_ms_p95": float(lat[int(0.95 * (len(lat) - 1))] * 1e3),
            "stream_chunks_decoded_again": redone,
            "final_nbest_mismatches_vs_cpu_on_sample": mism, "streams_checked": n_chk,
            "note": "stream_step(50 frames, host emissions) + prune(0) per chunk; value: chunks handed over back to back "
                    "(results read at the end), value_waiting_for_every_chunk / latencies: a synchronize after every "
                    "chunk; lexicon streams start "
                    "every chunk on the LDS-sized geometry and decode it again from the saved beam if a list overflows "
                    "(looked at when the next call needs the beam: the next chunk's upload runs under the kernel)"}


if __name__ == "__main__":
    main()
