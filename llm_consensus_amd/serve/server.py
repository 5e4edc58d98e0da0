"""OpenAI-compatible HTTP front door for an ensemble.

The reference's only interface is a stdin REPL (/root/reference/src/main.rs:428-471) that
serves one question at a time. A deployment needs a network endpoint and, more
importantly for throughput, a place where concurrent callers are merged into engine
batches. This module provides both with the standard library only:

    handler threads (ThreadingHTTPServer)  --jobs-->  one driver thread
                                                      collects up to ``max_batch`` jobs for
                                                      ``batch_window_ms``, answers them with
                                                      ONE ensemble.ask_many / backend.run,
                                                      wakes the handlers.

All GPU work happens on the driver thread, so engines never see concurrent callers.

Routes
    GET  /health                 liveness + queue depth
    GET  /v1/models              "consensus" plus every member model
    GET  /metrics                Prometheus text (server counters, engine stats)
    POST /v1/chat/completions    model "consensus" (default) runs the configured strategy on
                                 the last user message; any other model name is served by
                                 that member model directly (n, max_tokens, temperature,
                                 top_p, seed, stop honoured; finish_reason is "stop" or "length"). ``stream: true`` yields SSE chunks: one
                                 per engine step for member models (tokens as they are
                                 sampled), one for a consensus answer (it does not exist
                                 before the vote is in).
    POST /v1/completions         same for a raw prompt
    POST /v1/embeddings          pooled, L2-normalised embeddings from a member model
    POST /v1/score               extension: ``{"model", "prompt", "targets": [..]}`` -> the log-probability
                                 of each target continuation (its first token) after the prompt; with
                                 ``decision_targets`` on /v1/completions every choice carries the same for
                                 the position after its reply (``decision_logprobs``): the constrained
                                 decisions the strategies take (verdict words, choice letters), for callers
                                 that run their own voting
Consensus responses carry an extra ``consensus`` object (votes, rounds, agreement,
unanimity) next to the OpenAI fields.

With ``--api-key`` (or $CONSENSUS_API_KEY) the POST routes need ``Authorization: Bearer <key>``.
SIGTERM drains the server: new requests get 503, the admitted ones are answered, then it exits.

Malformed requests are answered, not crashed on: a bad ``Content-Length`` or a body that is not a
JSON object gets 400, an announced body over ``max_body_bytes`` gets 413 before any of it is read,
and socket reads and writes give up after ``io_timeout_s`` (half-sent requests, idle kept-alive
connections, stream readers that stall), so no client can hold a handler thread for ever. The
listen backlog is 1024; at most ``max_connections`` connections are open at a time (each has a
handler thread), one more is answered 503 from the accept loop.

At most ``max_pending`` requests are accepted at a time (answered or not yet): one more gets
429 with a Retry-After header instead of an unbounded wait behind engines that are already full.

A client that hangs up (closes its socket while it waits for an answer, or in the middle of a
token stream) is noticed within a quarter of a second; in continuous mode its work is cancelled:
the engines drop its requests at their next step boundary and their KV blocks return to the pool
(``cancelled_total`` in /metrics, ``requests.cancelled`` per engine). Members that are sharded
over a tensor / expert parallel group run such a request to its end.
"""
from __future__ import annotations

import json
import os
import queue
import select
import socket
import threading
import time
import uuid
from dataclasses import dataclass, field
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Any, Dict, List, Optional

from ..consensus.backend import BackendError, BackendStepper, GenParams
from ..utils.log import get_logger

log = get_logger("consensus.serve")
CONSENSUS_MODEL = "consensus"


@dataclass
class _Job:
    kind: str                      # consensus | generate | embed | score
    model: str
    text: str
    params: Optional[GenParams] = None
    targets: Optional[List[str]] = None      # score jobs: the continuations to score
    done: threading.Event = field(default_factory=threading.Event)
    # streaming requests: (chain, text) pieces as the engine produces them, None when finished
    chunks: Optional["queue.Queue"] = None
    result: Any = None
    error: Optional[str] = None
    cancelled: bool = False          # the client went away before its answer: stop working on it
    t_submit: float = field(default_factory=time.perf_counter)


class HTTPError(Exception):
    def __init__(self, status: int, message: str) -> None:
        super().__init__(message)
        self.status, self.message = status, message


class ConsensusServer:
    def __init__(self, ensemble, host: str = "127.0.0.1", port: int = 8000, max_batch: int = 64,
                 batch_window_ms: float = 5.0, request_timeout_s: float = 600.0, continuous: bool = True,
                 max_pending: int = 4096, api_key: Optional[str] = None, io_timeout_s: float = 60.0,
                 max_body_bytes: int = 16 << 20, max_connections: int = 8192) -> None:
        self.ensemble = ensemble
        self.continuous = continuous
        self.max_batch, self.window = max_batch, batch_window_ms / 1e3
        self.request_timeout_s = request_timeout_s
        # Admission limit: requests accepted and not yet answered. One more is refused with 429 and a
        # Retry-After instead of queueing without bound behind work that already fills the engines.
        self.max_pending = max_pending
        self._outstanding = 0
        self.api_key = api_key or None           # when set, POST routes need "Authorization: Bearer <key>"
        self.max_body_bytes = max_body_bytes     # a request that announces more gets 413 before a byte of it is read
        self.jobs: "queue.Queue[_Job]" = queue.Queue()
        self.cancels: "queue.Queue[_Job]" = queue.Queue()      # jobs whose client disconnected (see _cancel)
        self._cancels_served = False     # only the single-process continuous driver takes work out of engines
        self.counters: Dict[str, float] = {"requests_total": 0, "errors_total": 0, "batches_total": 0,
                                           "batched_requests_total": 0, "latency_seconds_sum": 0.0,
                                           "cancelled_total": 0, "rejected_total": 0}
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self._draining = threading.Event()       # set by drain(): refuse new requests, finish the admitted ones
        server = self
        max_connections_ = max_connections

        class Handler(BaseHTTPRequestHandler):
            protocol_version = "HTTP/1.1"
            # Socket reads and writes of a connection give up after this long: a client that opens a
            # connection and sends nothing (or half a request, or stops reading its token stream)
            # gives its handler thread back instead of holding it for ever. The wait for an ANSWER
            # is not a socket operation and has its own limit (request_timeout_s).
            timeout = io_timeout_s

            def log_message(self, fmt, *args):  # route access logs through our logger
                log.debug("%s " + fmt, self.address_string(), *args)

            def do_GET(self):
                server._dispatch(self, "GET")

            def do_POST(self):
                server._dispatch(self, "POST")

        class Server(ThreadingHTTPServer):
            # socketserver listens with a backlog of 5: a few hundred clients that connect at the same
            # moment (every connection of a closed loop after a step that finished them all) overflow
            # it, and the kernel then drops or resets connections (1 s stalls, ECONNRESET) although
            # the server is idle enough to take them (measured with benchmarks/http_load.py).
            request_queue_size = 1024
            # One handler thread per open connection: beyond this many, a new connection is told 503
            # from the accept loop and closed instead of getting a thread of its own.
            max_connections = max_connections_
            active = 0
            active_lock = threading.Lock()

            def process_request(self, request, client_address):
                with self.active_lock:
                    over = self.active >= self.max_connections
                    if not over:
                        self.active += 1
                if over:
                    with server._lock:
                        server.counters["rejected_total"] += 1
                    try:
                        request.settimeout(0.2)
                        request.sendall(b"HTTP/1.1 503 Service Unavailable\r\nRetry-After: 1\r\nConnection: close\r\n"
                                        b"Content-Type: application/json\r\nContent-Length: 72\r\n\r\n"
                                        b'{"error": {"message": "too many open connections", "code": 503}}        ')
                    except OSError:
                        pass
                    self.shutdown_request(request)
                    return
                super().process_request(request, client_address)

            def process_request_thread(self, request, client_address):
                try:
                    super().process_request_thread(request, client_address)
                finally:
                    with self.active_lock:
                        self.active -= 1

            def handle_error(self, request, client_address):
                # a client that resets its connection is not a server error worth a traceback
                import sys

                e = sys.exc_info()[1]
                if isinstance(e, (ConnectionResetError, BrokenPipeError, TimeoutError)):
                    log.debug("client %s went away: %s", client_address, e)
                else:
                    super().handle_error(request, client_address)

        self.httpd = Server((host, port), Handler)
        self.httpd.daemon_threads = True
        self.host, self.port = self.httpd.server_address[:2]
        self._driver = threading.Thread(target=self._drive, name="consensus-driver", daemon=True)
        self._http = threading.Thread(target=self.httpd.serve_forever, name="consensus-http", daemon=True)

    # ------------------------------------------------------------------ lifecycle
    def start(self) -> "ConsensusServer":
        self._driver.start()
        self._http.start()
        log.info("serving on http://%s:%d", self.host, self.port)
        return self

    def serve_forever(self, drain_grace_s: float = 30.0) -> None:
        """Runs the engine driver on the calling thread (GPU contexts stay on the main
        thread) and the HTTP accept loop on a helper thread. SIGTERM (and a first Ctrl-C) drain
        the server: new requests are refused with 503, the ones in flight are answered (for at
        most ``drain_grace_s`` seconds), then the loop ends."""
        import signal

        self._http.start()
        log.info("serving on http://%s:%d", self.host, self.port)
        old = {}
        if threading.current_thread() is threading.main_thread():
            for sig in (signal.SIGTERM, signal.SIGINT):
                old[sig] = signal.signal(sig, lambda *_: self.drain(drain_grace_s))
        try:
            self._drive()
        except KeyboardInterrupt:
            pass
        finally:
            for sig, h in old.items():
                signal.signal(sig, h)
            self.shutdown()

    def drain(self, grace_s: float = 30.0) -> None:
        """Stops admitting requests and ends the serving loop once the admitted ones are answered
        (or ``grace_s`` has passed). Returns at once; a second call stops the loop immediately."""
        if self._draining.is_set():
            self._stop.set()
            return
        self._draining.set()
        log.info("draining: %d request(s) in flight", self._outstanding)

        def wait():
            deadline = time.perf_counter() + grace_s
            while self._outstanding > 0 and time.perf_counter() < deadline and not self._stop.is_set():
                time.sleep(0.05)
            self._stop.set()

        threading.Thread(target=wait, daemon=True).start()

    def shutdown(self) -> None:
        self._stop.set()
        self.httpd.shutdown()
        self.httpd.server_close()
        if self._driver.is_alive():
            self._driver.join(timeout=5)

    # ------------------------------------------------------------------ driver
    def _collect(self) -> List[_Job]:
        try:
            first = self.jobs.get(timeout=0.05)
        except queue.Empty:
            return []
        batch = [first]
        deadline = time.perf_counter() + self.window
        while len(batch) < self.max_batch:
            left = deadline - time.perf_counter()
            try:
                batch.append(self.jobs.get(timeout=max(left, 0)) if left > 0 else self.jobs.get_nowait())
            except queue.Empty:
                break
        return batch

    def _continuous_ok(self) -> bool:
        """Strategies that offer a pool (the reference protocol, majority vote) over backends
        that live in this process are driven without batch boundaries; everything else runs
        batch by batch."""
        strat = getattr(self.ensemble, "strategy", None)
        return (self.continuous and callable(getattr(strat, "pool", None))
                and all(getattr(b, "incremental", False) for b in self.ensemble.backends.values()))

    def _drive_continuous(self) -> None:
        """New questions join the sessions in flight at the next engine step; every request is
        answered as soon as its own session (or generation) is done."""
        # ONE stepper for every backend of the ensemble, shared with the pool: an engine must never
        # be stepped from two drivers (a step the pool's driver has in flight would be followed by
        # a second schedule() from a direct poll before its results were applied).
        stepper = BackendStepper(list({id(b): b for b in self.ensemble.backends.values()}.values()))
        self._cancels_served = True

        def new_pool():
            p = self.ensemble.strategy.pool()
            p.stepper = stepper
            return p

        pool = new_pool()
        tickets: Dict[int, _Job] = {}
        direct: List[tuple] = []
        while not self._stop.is_set():
            # (an engine that still holds a cancelled request is not idle: it lets go at its next step)
            idle = not tickets and not direct and not any(b.has_leftovers() for b in stepper.backends)
            try:
                new = [self.jobs.get(timeout=0.05)] if idle else []
            except queue.Empty:
                continue
            while len(tickets) + len(direct) + len(new) < 4 * self.max_batch:
                try:
                    new.append(self.jobs.get_nowait())
                except queue.Empty:
                    break
            if new:
                with self._lock:
                    self.counters["batches_total"] += 1
                    self.counters["batched_requests_total"] += len(new)
            while True:                     # clients that went away: their work leaves the engines
                try:
                    gone = self.cancels.get_nowait()
                except queue.Empty:
                    break
                for t, j in list(tickets.items()):
                    if j is gone:
                        pool.cancel(t)
                        del tickets[t]
                        j.error = "cancelled"
                        self._finish(j)
                for h_, j in direct:
                    if j is gone:
                        h_.cancel()         # completes with an error at the engine's next step
            for j in new:
                if j.cancelled:
                    j.error = "cancelled"
                    self._finish(j)
                    continue
                try:
                    if j.kind == "consensus":
                        tickets[pool.submit(j.text)] = j
                    else:
                        be = self.ensemble.backends[j.model]
                        direct.append((self._submit_direct(be, j), j))
                except Exception as e:  # noqa: BLE001
                    j.error = f"{type(e).__name__}: {e}"
                    self._finish(j)
            try:
                was_active = pool.active
                finished = pool.pump()
                # The pool advances the shared stepper while it has questions; when it is idle the
                # direct requests are advanced here, through the same stepper.
                # ... and so are the leftovers of cancelled requests: an engine lets them go at its
                # next step boundary, which has to come even if nobody else waits for that engine.
                leftovers = not was_active and not direct and any(b.has_leftovers() for b in stepper.backends)
                if not was_active and (direct or leftovers):
                    if not stepper.step():
                        time.sleep(20e-6)
            except Exception as e:  # noqa: BLE001  never let the driver die with waiters parked on it
                log.error("engine step failed: %s", e)
                for j in list(tickets.values()) + [j for _, j in direct]:
                    j.error = f"{type(e).__name__}: {e}"
                    self._finish(j)
                tickets.clear()
                direct.clear()
                pool = new_pool()
                continue
            for t, res in finished:
                j = tickets.pop(t)
                j.result = res
                self.ensemble.record([res])
                j.done.set()
            still = []
            for h, j in direct:
                if not h.done:
                    still.append((h, j))
                    continue
                try:
                    j.result = h.result()
                except BackendError as e:
                    j.error = str(e)
                self._finish(j)
            direct = still

    def _submit_direct(self, be, j: _Job):
        """A request addressed to one member model goes straight to its backend."""
        if j.kind == "embed":
            return be.submit_embed(j.text)
        if j.kind == "score":
            return be.submit_score(j.text, j.targets)
        return be.submit_generate(j.text, j.params, self._on_text(j))

    @staticmethod
    def _on_text(j: _Job):
        if j.chunks is None:
            return None
        return lambda chain, text: j.chunks.put((chain, text))

    @staticmethod
    def _finish(j: _Job) -> None:
        j.done.set()
        if j.chunks is not None:
            j.chunks.put(None)

    def _drive_lockstep(self) -> None:
        """An ensemble that spans several ranks (parallel.ensemble_parallel.LockstepEnsemble): every
        pass of this loop is one lock-step round of the whole group. New questions travel to the
        other ranks in the round's exchange and join the sessions in flight; answers come back in
        the exchange of the round after their last member call. While nothing is in flight a
        round is the group's heartbeat."""
        ens = self.ensemble
        tickets: Dict[int, _Job] = {}
        while not self._stop.is_set():
            new: List[_Job] = []
            if not tickets:
                try:
                    new.append(self.jobs.get(timeout=0.05))
                except queue.Empty:
                    pass
            while len(tickets) + len(new) < 4 * self.max_batch:
                try:
                    new.append(self.jobs.get_nowait())
                except queue.Empty:
                    break
            if new:
                with self._lock:
                    self.counters["batches_total"] += 1
                    self.counters["batched_requests_total"] += len(new)
            for j in new:
                if j.kind != "consensus":
                    j.error = "member models are not individually addressable when the ensemble spans several ranks"
                    self._finish(j)
                else:
                    tickets[ens.submit(j.text)] = j
            try:
                finished = ens.poll()
            except Exception as e:  # noqa: BLE001  never let the driver die with waiters parked on it
                log.error("lock-step round failed: %s", e)
                for j in tickets.values():
                    j.error = f"{type(e).__name__}: {e}"
                    self._finish(j)
                raise
            for t, res in finished:
                j = tickets.pop(t, None)
                if j is not None:
                    j.result = res
                    j.done.set()

    def _drive(self) -> None:
        if callable(getattr(self.ensemble, "poll", None)) and callable(getattr(self.ensemble, "submit", None)):
            return self._drive_lockstep()
        if self._continuous_ok():
            return self._drive_continuous()
        while not self._stop.is_set():
            batch = self._collect()
            if not batch:
                continue
            with self._lock:
                self.counters["batches_total"] += 1
                self.counters["batched_requests_total"] += len(batch)
            try:
                self._run_batch(batch)
            except Exception as e:  # never let the driver die with waiters parked on it
                log.error("batch failed: %s", e)
                for j in batch:
                    if not j.done.is_set():
                        j.error = f"{type(e).__name__}: {e}"
            for j in batch:
                self._finish(j)

    def _run_batch(self, batch: List[_Job]) -> None:
        cons = [j for j in batch if j.kind == "consensus"]
        if cons:
            for j, res in zip(cons, self.ensemble.ask_many([j.text for j in cons])):
                j.result = res
        direct = [j for j in batch if j.kind != "consensus"]
        handles = []
        for j in direct:
            handles.append(self._submit_direct(self.ensemble.backends[j.model], j))
        streaming = any(j.chunks is not None for j in direct)
        for name in {j.model for j in direct}:
            be = self.ensemble.backends[name]
            if streaming and getattr(be, "incremental", False):
                while be.poll():            # step by step, so streamed tokens leave as they are sampled
                    pass
            else:
                be.run()
        for j, h in zip(direct, handles):
            try:
                j.result = h.result()
            except BackendError as e:
                j.error = str(e)

    def _cancel(self, job: _Job) -> None:
        """The client of ``job`` is gone: the serving loop takes its work out of the engines (KV
        blocks freed at the next step) instead of finishing an answer nobody will read. The
        reference cannot stop a member call once it is sent (/root/reference/src/main.rs:82-86)."""
        if job.cancelled or job.done.is_set():
            return
        job.cancelled = True
        with self._lock:
            self.counters["cancelled_total"] += 1
        if self._cancels_served:
            self.cancels.put(job)

    @staticmethod
    def _peer_closed(conn) -> bool:
        """True once the client has closed its end (end of stream readable on the socket)."""
        try:
            r, _, _ = select.select([conn], [], [], 0)
            if not r:
                return False
            return conn.recv(1, socket.MSG_PEEK | socket.MSG_DONTWAIT) == b""
        except (BlockingIOError, InterruptedError):
            return False
        except (OSError, ValueError):
            return True

    def submit(self, job: _Job, conn=None) -> _Job:
        """Queues ``job`` and waits for it. With ``conn`` (the request's socket) the wait also
        watches for the client hanging up, and cancels the job if it does."""
        self.jobs.put(job)
        deadline = time.perf_counter() + self.request_timeout_s
        while not job.done.wait(0.25 if conn is not None else self.request_timeout_s):
            if conn is not None and self._peer_closed(conn):
                self._cancel(job)
                raise HTTPError(499, "client closed the connection")
            if time.perf_counter() >= deadline:
                self._cancel(job)           # nobody waits for it any more
                raise HTTPError(504, "request timed out in the engine queue")
        if job.error:
            raise HTTPError(500, job.error)
        return job

    # ------------------------------------------------------------------ http
    def _dispatch(self, h: BaseHTTPRequestHandler, method: str) -> None:
        t0 = time.perf_counter()
        with self._lock:
            self.counters["requests_total"] += 1
        try:
            path = h.path.split("?", 1)[0].rstrip("/") or "/"
            if method == "GET":
                if path in ("/health", "/healthz"):
                    return self._send_json(h, 200, {"status": "ok", "queued": self.jobs.qsize()})
                if path == "/v1/models":
                    return self._send_json(h, 200, self._models())
                if path == "/metrics":
                    return self._send(h, 200, self._metrics().encode(), "text/plain; version=0.0.4")
                raise HTTPError(404, f"no route {path}")
            try:
                n = int(h.headers.get("Content-Length") or 0)
            except ValueError:
                n = -1
            if n < 0:
                h.close_connection = True        # the body's length is unknown: the connection cannot be reused
                raise HTTPError(400, "bad Content-Length")
            if n > self.max_body_bytes:
                h.close_connection = True
                raise HTTPError(413, f"request body of {n} bytes; at most {self.max_body_bytes} are accepted")
            try:
                body = json.loads(h.rfile.read(n) or b"{}")
            except ValueError as e:              # not JSON, or not UTF-8
                raise HTTPError(400, f"invalid JSON body: {e}")
            if not isinstance(body, dict):
                raise HTTPError(400, "the request body must be a JSON object")
            if path not in ("/v1/chat/completions", "/v1/completions", "/v1/embeddings", "/v1/score"):
                raise HTTPError(404, f"no route {path}")
            if self.api_key is not None:
                import hmac

                given = (h.headers.get("Authorization") or "").removeprefix("Bearer ").strip()
                if not hmac.compare_digest(given.encode(), self.api_key.encode()):
                    raise HTTPError(401, "missing or wrong API key (Authorization: Bearer ...)")
            if self._draining.is_set():
                raise HTTPError(503, "the server is shutting down")
            with self._lock:
                admitted = self._outstanding < self.max_pending
                if admitted:
                    self._outstanding += 1
                else:
                    self.counters["rejected_total"] += 1
            if not admitted:
                raise HTTPError(429, f"{self.max_pending} requests are already in flight; retry later")
            try:
                if path == "/v1/chat/completions":
                    return self._completion(h, body, chat=True)
                if path == "/v1/completions":
                    return self._completion(h, body, chat=False)
                if path == "/v1/score":
                    return self._score(h, body)
                return self._embeddings(h, body)
            finally:
                with self._lock:
                    self._outstanding -= 1
        except HTTPError as e:
            with self._lock:
                self.counters["errors_total"] += 1
            try:
                self._send_json(h, e.status, {"error": {"message": e.message, "type": "invalid_request_error"
                                                        if e.status < 500 else "server_error", "code": e.status}})
            except (BrokenPipeError, ConnectionResetError):
                pass                     # the client this error is for has gone (status 499 always ends here)
        except OSError:
            h.close_connection = True            # the client went away, or stopped sending / reading
        except Exception as e:  # noqa: BLE001  a bug in a handler must not cost the client its answer line
            log.error("request %s %s failed: %s: %s", method, h.path, type(e).__name__, e, exc_info=e)
            with self._lock:
                self.counters["errors_total"] += 1
            h.close_connection = True
            try:
                self._send_json(h, 500, {"error": {"message": f"{type(e).__name__}: {e}", "type": "server_error", "code": 500}})
            except OSError:
                pass
        finally:
            with self._lock:
                self.counters["latency_seconds_sum"] += time.perf_counter() - t0

    def _send(self, h, status: int, payload: bytes, ctype: str) -> None:
        h.send_response(status)
        if status in (429, 503):
            h.send_header("Retry-After", "1")
        h.send_header("Content-Type", ctype)
        h.send_header("Content-Length", str(len(payload)))
        h.end_headers()
        h.wfile.write(payload)

    def _send_json(self, h, status: int, obj: Any) -> None:
        self._send(h, status, json.dumps(obj).encode(), "application/json")

    def _models(self) -> Dict[str, Any]:
        now = int(time.time())
        ids = [CONSENSUS_MODEL] + sorted(self.ensemble.backends)
        return {"object": "list",
                "data": [{"id": i, "object": "model", "created": now, "owned_by": "llm_consensus_amd"} for i in ids]}

    def _metrics(self) -> str:
        # Built at most twice a second (scrapers ask every few seconds): collecting every engine's
        # statistics holds the interpreter lock, which the thread that drives the engines needs too.
        # This bounds the cost of the collection only: every request, however cheap, costs handler
        # time under the same lock, so a flood of them still slows serving (hundreds of GETs per
        # second halve a CPU-backed run); rate limiting belongs to the proxy in front.
        now = time.perf_counter()
        cached = getattr(self, "_metrics_cache", None)
        if cached is not None and now - cached[0] < 0.5:
            return cached[1]
        text = self._metrics_text()
        self._metrics_cache = (now, text)
        return text

    def _metrics_text(self) -> str:
        lines = []
        with self._lock:
            for k, v in self.counters.items():
                lines.append(f"# TYPE consensus_http_{k} counter")
                lines.append(f"consensus_http_{k} {v}")
        lines.append(f"consensus_http_queue_depth {self.jobs.qsize()}")
        lines.append(f"consensus_http_requests_in_flight {self._outstanding}")
        for name, st in self.ensemble.stats().items():
            for k, v in st.items():
                if isinstance(v, (int, float)):
                    lines.append(f'consensus_engine_{k}{{model="{name}"}} {v}')
        return "\n".join(lines) + "\n"

    # ---- request parsing
    def _gen_params(self, body: Dict[str, Any]) -> GenParams:
        s = self.ensemble.config.sampling
        mt = body.get("max_completion_tokens", body.get("max_tokens"))
        stop = body.get("stop")
        if isinstance(stop, str):
            stop = [stop]
        if stop is not None and (not isinstance(stop, list) or len(stop) > 4
                                 or not all(isinstance(t, str) and t for t in stop)):
            raise HTTPError(400, "'stop' must be a non-empty string or a list of up to 4 of them")
        dt = body.get("decision_targets")
        if dt is not None and (not isinstance(dt, list) or not dt or len(dt) > 64
                               or not all(isinstance(t, str) and t for t in dt)):
            raise HTTPError(400, "'decision_targets' must be a list of 1-64 non-empty strings")
        def number(name, default, lo, hi, integer=False, lo_open=False):
            v = body.get(name)
            if v is None:
                return default
            ok = isinstance(v, int) and not isinstance(v, bool) if integer else \
                isinstance(v, (int, float)) and not isinstance(v, bool)
            if not ok or v != v or v > hi or v < lo or (lo_open and v == lo):
                kind = "an integer" if integer else "a number"
                raise HTTPError(400, f"'{name}' must be {kind} in {'(' if lo_open else '['}{lo}, {hi}], not {v!r}")
            return v

        if mt is not None and (not isinstance(mt, int) or isinstance(mt, bool)):
            raise HTTPError(400, f"'max_tokens' must be an integer, not {mt!r}")
        return GenParams(
            decision_targets=dt,
            stop=stop or None,
            temperature=float(number("temperature", s.temperature, 0, 100)),
            top_p=float(number("top_p", s.top_p, 0, 1, lo_open=True)),
            top_k=int(number("top_k", s.top_k, 0, 2 ** 31 - 1, integer=True)),
            max_new_tokens=int(mt if mt is not None else s.max_new_tokens),
            seed=int(number("seed", s.seed, 0, 2 ** 62, integer=True)),
            n=int(number("n", 1, 1, 64, integer=True)),
        )

    @staticmethod
    def _chat_text(messages: List[Dict[str, Any]], consensus: bool) -> str:
        if not isinstance(messages, list) or not messages:
            raise HTTPError(400, "'messages' must be a non-empty list")

        if not all(isinstance(m, dict) and isinstance(m.get("role", "user"), str) for m in messages):
            raise HTTPError(400, "every message must be an object with a string 'role'")

        def content(m):
            c = m.get("content", "")
            if isinstance(c, list):  # content parts
                if not all(isinstance(p, dict) and isinstance(p.get("text", ""), str) for p in c):
                    raise HTTPError(400, "content parts must be objects with a string 'text'")
                c = "".join(p.get("text", "") for p in c)
            if c is None:
                c = ""
            if not isinstance(c, str):
                raise HTTPError(400, "'content' must be a string or a list of text parts")
            return c

        if consensus:  # the strategy wraps the question in its own prompts
            users = [m for m in messages if m.get("role") == "user"]
            if not users:
                raise HTTPError(400, "no user message")
            return content(users[-1])
        return "\n".join(f"{m.get('role', 'user')}: {content(m)}" for m in messages) + "\nassistant:"

    def _completion(self, h, body: Dict[str, Any], chat: bool) -> None:
        model = body.get("model") or CONSENSUS_MODEL
        if not isinstance(model, str):
            raise HTTPError(400, "'model' must be a string")
        consensus = model == CONSENSUS_MODEL
        if not consensus and model not in self.ensemble.backends:
            raise HTTPError(404, f"unknown model {model!r}")
        if chat:
            text = self._chat_text(body.get("messages"), consensus)
        else:
            text = body.get("prompt")
            if isinstance(text, list):
                if len(text) != 1:
                    raise HTTPError(400, "one prompt per request (send concurrent requests; they are batched)")
                text = text[0]
            if not isinstance(text, str) or not text:
                raise HTTPError(400, "'prompt' must be a non-empty string")
        if body.get("stream") is not None and not isinstance(body.get("stream"), bool):
            raise HTTPError(400, "'stream' must be true or false")
        params = self._gen_params(body)
        if params.max_new_tokens < 1 or params.n < 1:
            raise HTTPError(400, "'max_tokens' and 'n' must be at least 1")
        if not consensus:
            limit = getattr(getattr(self.ensemble.backends[model], "engine", None), "max_model_len", None)
            if limit is not None and params.max_new_tokens >= limit:
                raise HTTPError(400, f"'max_tokens' = {params.max_new_tokens} leaves no room for a prompt in "
                                     f"{model}'s context of {limit} tokens")
        rid = ("chatcmpl-" if chat else "cmpl-") + uuid.uuid4().hex[:24]
        created = int(time.time())
        if params.decision_targets and (consensus or body.get("stream")):
            raise HTTPError(400, "'decision_targets' goes with a member model and a whole (not streamed) reply")
        if body.get("stream") and not consensus:
            return self._stream_tokens(h, rid, created, model, text, params, chat)
        job = self.submit(_Job("consensus" if consensus else "generate", model, text, params), h.connection)
        extra: Dict[str, Any] = {}
        if consensus:
            r = job.result
            texts = [r.answer]
            u = getattr(r, "usage", None) or {}
            usage = {"prompt_tokens": int(u.get("prompt_tokens", 0)), "completion_tokens": int(u.get("completion_tokens", 0)),
                     "total_tokens": int(u.get("total_tokens", 0)),
                     "prompt_tokens_details": {"cached_tokens": int(u.get("cached_tokens", 0))}}
            extra["consensus"] = {"strategy": r.strategy, "rounds": r.rounds, "member_calls": r.member_calls,
                                  "unanimous": r.unanimous, "forced": r.forced, "agreement": r.agreement,
                                  "proposer": r.proposer, "latency_s": r.latency_s, "votes": r.votes}
        else:
            g = job.result
            texts = g.texts
            ct = sum(len(t) for t in g.token_ids)
            usage = {"prompt_tokens": g.prompt_tokens, "completion_tokens": ct,
                     "total_tokens": g.prompt_tokens + ct,
                     "prompt_tokens_details": {"cached_tokens": g.cached_tokens}}
            if g.decision_logprobs:
                extra["_decision"] = g.decision_logprobs
        fr = list(getattr(job.result, "finish_reasons", None) or [])
        reasons = [fr[i] if i < len(fr) else "stop" for i in range(len(texts))]
        decision = extra.pop("_decision", None)
        if body.get("stream"):
            return self._stream(h, rid, created, model, texts, chat, extra)
        if chat:
            choices = [{"index": i, "message": {"role": "assistant", "content": t}, "finish_reason": reasons[i]}
                       for i, t in enumerate(texts)]
            obj = "chat.completion"
        else:
            choices = [{"index": i, "text": t, "finish_reason": reasons[i], "logprobs": None} for i, t in enumerate(texts)]
            obj = "text_completion"
        if decision:
            for c, d in zip(choices, decision):
                c["decision_logprobs"] = [float(x) for x in d]
        self._send_json(h, 200, {"id": rid, "object": obj, "created": created, "model": model,
                                 "choices": choices, "usage": usage, **extra})

    def _stream_tokens(self, h, rid, created, model, text, params, chat) -> None:
        """Server-sent events of a member model's completion: the handler thread relays the
        pieces of text the driver thread's engine steps produce (one chunk per step and chain),
        then the stop chunks with the usage record."""
        job = _Job("generate", model, text, params, chunks=queue.Queue())
        self.jobs.put(job)
        h.send_response(200)
        h.send_header("Content-Type", "text/event-stream")
        h.send_header("Cache-Control", "no-cache")
        h.send_header("Connection", "close")
        h.end_headers()
        obj = "chat.completion.chunk" if chat else "text_completion"

        def emit(choice, **more):
            payload = {"id": rid, "object": obj, "created": created, "model": model, "choices": [choice], **more}
            h.wfile.write(b"data: " + json.dumps(payload).encode() + b"\n\n")
            h.wfile.flush()

        started = set()
        deadline = time.perf_counter() + self.request_timeout_s
        try:
            self._stream_tokens_body(h, job, emit, started, deadline, chat)
        except OSError:                 # reset, broken pipe, or a reader so slow that a write timed out
            self._cancel(job)           # nobody reads the stream any more: stop generating
            raise

    def _stream_tokens_body(self, h, job, emit, started, deadline, chat) -> None:
        while True:
            try:
                item = job.chunks.get(timeout=min(0.25, max(0.0, deadline - time.perf_counter())))
            except queue.Empty:
                if time.perf_counter() < deadline:
                    if self._peer_closed(h.connection):     # hung up between two tokens
                        raise BrokenPipeError("client closed the connection")
                    continue
                self._cancel(job)
                emit({"index": 0, "delta": {}, "finish_reason": "error"}, error={"message": "request timed out"})
                break
            if item is None:
                break
            i, piece = item
            if chat:
                delta = {"content": piece}
                if i not in started:
                    delta["role"] = "assistant"
                    started.add(i)
                emit({"index": i, "delta": delta, "finish_reason": None})
            else:
                emit({"index": i, "text": piece, "finish_reason": None})
        if job.error:
            emit({"index": 0, "delta": {}, "finish_reason": "error"}, error={"message": job.error})
        elif job.result is not None:
            g = job.result
            ct = sum(len(t) for t in g.token_ids)
            usage = {"prompt_tokens": g.prompt_tokens, "completion_tokens": ct, "total_tokens": g.prompt_tokens + ct}
            fr = list(getattr(g, "finish_reasons", None) or [])
            for i in range(len(g.texts)):
                last = {"usage": usage} if i == len(g.texts) - 1 else {}
                why = fr[i] if i < len(fr) else "stop"
                emit({"index": i, "delta": {}, "finish_reason": why} if chat
                     else {"index": i, "text": "", "finish_reason": why}, **last)
        h.wfile.write(b"data: [DONE]\n\n")
        h.wfile.flush()
        h.close_connection = True

    def _stream(self, h, rid, created, model, texts, chat, extra) -> None:
        """Server-sent events for answers that are produced whole (a consensus answer does not
        exist until the vote is in): each choice is one content chunk followed by a stop chunk."""
        h.send_response(200)
        h.send_header("Content-Type", "text/event-stream")
        h.send_header("Cache-Control", "no-cache")
        h.send_header("Connection", "close")
        h.end_headers()
        obj = "chat.completion.chunk" if chat else "text_completion"

        def emit(choice, **more):
            payload = {"id": rid, "object": obj, "created": created, "model": model, "choices": [choice], **more}
            h.wfile.write(b"data: " + json.dumps(payload).encode() + b"\n\n")

        for i, t in enumerate(texts):
            if chat:
                emit({"index": i, "delta": {"role": "assistant", "content": t}, "finish_reason": None})
                emit({"index": i, "delta": {}, "finish_reason": "stop"}, **extra)
            else:
                emit({"index": i, "text": t, "finish_reason": "stop"}, **extra)
        h.wfile.write(b"data: [DONE]\n\n")
        h.wfile.flush()
        h.close_connection = True

    def _score(self, h, body: Dict[str, Any]) -> None:
        """Log-probability of each target's first token after the prompt, from one member model: the
        constrained decision the strategies take on the device (verdict words, choice letters,
        /root/reference/src/main.rs:144-148 parses them out of free text instead)."""
        model = body.get("model")
        if not isinstance(model, str):
            raise HTTPError(400, "'model' must be a string")
        if model not in self.ensemble.backends:
            raise HTTPError(404, f"unknown model {model!r}")
        prompt, targets = body.get("prompt"), body.get("targets")
        if not isinstance(prompt, str) or not prompt:
            raise HTTPError(400, "'prompt' must be a non-empty string")
        if (not isinstance(targets, list) or not targets or len(targets) > 64
                or not all(isinstance(t, str) and t for t in targets)):
            raise HTTPError(400, "'targets' must be a list of 1-64 non-empty strings")
        job = self.submit(_Job("score", model, prompt, targets=list(targets)), h.connection)
        self._send_json(h, 200, {"object": "score", "model": model, "targets": targets,
                                 "logprobs": [float(x) for x in job.result]})

    def _embeddings(self, h, body: Dict[str, Any]) -> None:
        model = body.get("model")
        if model is not None and not isinstance(model, str):
            raise HTTPError(400, "'model' must be a string")
        if not model or model == CONSENSUS_MODEL:
            emb = [m for m in self.ensemble.members if m.role == "embedder"] or self.ensemble.members
            model = next(k for k, b in self.ensemble.backends.items() if b is emb[0].backend)
        if model not in self.ensemble.backends:
            raise HTTPError(404, f"unknown model {model!r}")
        inp = body.get("input")
        inputs = [inp] if isinstance(inp, str) else inp
        if not isinstance(inputs, list) or not inputs or not all(isinstance(x, str) and x for x in inputs):
            raise HTTPError(400, "'input' must be a non-empty string or list of strings")
        if len(inputs) > 256:
            raise HTTPError(400, "at most 256 inputs per embeddings request")
        jobs = [_Job("embed", model, x) for x in inputs]
        for j in jobs[:-1]:
            self.jobs.put(j)
        try:
            self.submit(jobs[-1], h.connection)
            data = []
            for i, j in enumerate(jobs):
                if not j.done.wait(self.request_timeout_s):
                    raise HTTPError(504, "request timed out in the engine queue")
                if j.error:
                    raise HTTPError(500, j.error)
                data.append({"object": "embedding", "index": i, "embedding": [float(x) for x in j.result]})
        except HTTPError:
            for j in jobs:              # the caller is gone or out of time: none of its inputs is worth computing
                self._cancel(j)
            raise
        self._send_json(h, 200, {"object": "list", "data": data, "model": model,
                                 "usage": {"prompt_tokens": 0, "total_tokens": 0}})


def main(argv: Optional[list] = None) -> int:
    import argparse

    from ..app import build_ensemble
    from ..config import load_config_or_exit

    ap = argparse.ArgumentParser(prog="llm-consensus-serve", description="OpenAI-compatible consensus endpoint")
    ap.add_argument("--config")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--max-batch", type=int, default=64)
    ap.add_argument("--batch-window-ms", type=float, default=5.0)
    ap.add_argument("--api-key", default=os.environ.get("CONSENSUS_API_KEY"),
                    help="require 'Authorization: Bearer <key>' on the POST routes (default: $CONSENSUS_API_KEY, unset = open)")
    ap.add_argument("--request-timeout", type=float, default=600.0,
                    help="seconds a request may take before it is answered 504 and its work is cancelled")
    ap.add_argument("--max-pending", type=int, default=4096,
                    help="requests accepted and not yet answered; one more gets 429 with Retry-After")
    ap.add_argument("--io-timeout", type=float, default=60.0,
                    help="seconds a socket read or write may take (a client that sends half a request, idles on a "
                         "kept-alive connection or stops reading its stream gives its handler thread back)")
    ap.add_argument("--max-body-mb", type=float, default=16.0, help="largest request body accepted (413 beyond)")
    ap.add_argument("--max-connections", type=int, default=8192,
                    help="open connections served at a time (each has a handler thread); one more is told 503")
    ap.add_argument("--gpus", type=int, default=None,
                    help="run as this many ranks, one per GPU, started by this command (default: every replica "
                         "of the configured ensemble that the visible GPUs hold)")
    ap.add_argument("--device", help="cuda or cpu (default: GPU when present)")
    ap.add_argument("--distributed", action="store_true",
                    help="this process is ONE rank of a group started by a launcher (this command with --gpus, "
                         "or torch.distributed.run); rank 0 serves HTTP")
    ap.add_argument("--no-warm-up", action="store_true",
                    help="do not record the decode graphs before the socket opens (the first requests of every "
                         "batch shape then pay for it)")
    ap.add_argument("--batched", action="store_true",
                    help="answer in closed batches even where the continuous driver applies")
    import sys

    argv = sys.argv[1:] if argv is None else list(argv)
    args = ap.parse_args(argv)
    cfg = load_config_or_exit(args.config, "llm-consensus-serve")
    if args.port and int(os.environ.get("RANK", "0")) == 0:
        # an address that cannot be bound is found out now, not after minutes of model loading
        probe = socket.socket(socket.AF_INET6 if ":" in args.host else socket.AF_INET)
        try:
            probe.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            probe.bind((args.host, args.port))
        except OSError as e:
            print(f"llm-consensus-serve: error: cannot listen on {args.host}:{args.port}: {e}", file=sys.stderr, flush=True)
            return 2
        finally:
            probe.close()
    from ..parallel import launch

    if not args.distributed and not launch.under_launcher():
        n = launch.ranks_for(cfg, args.gpus, args.device, replicate=True)
        if n > 1:
            return launch.launch_ranks("llm_consensus_amd.serve.server",
                                       launch.strip_flag(argv, "--gpus") + ["--distributed"], n)
        packed = launch.warn_if_packed(cfg, n, args.device)
        if packed:
            print(packed, file=sys.stderr, flush=True)
    elif launch.under_launcher():
        args.distributed = True
    if args.distributed:
        from ..parallel.ensemble_parallel import init_distributed, serve_distributed

        comm, device = init_distributed(args.device)
        ens = serve_distributed(cfg, comm, device, continuous=not args.batched)
        if ens is None:
            return 0
    else:
        import torch

        ens = build_ensemble(cfg, torch.device(args.device) if args.device else None)
    if not args.no_warm_up:
        ens.warm_up()
    try:
        ConsensusServer(ens, args.host, args.port, args.max_batch, args.batch_window_ms,
                        request_timeout_s=args.request_timeout, continuous=not args.batched,
                        max_pending=args.max_pending, api_key=args.api_key, io_timeout_s=args.io_timeout,
                        max_body_bytes=int(args.max_body_mb * (1 << 20)),
                        max_connections=args.max_connections).serve_forever()
    finally:
        if args.distributed:
            ens.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
