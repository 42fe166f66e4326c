"""HTTP shell around the GPU SLIM model: the drop-in for rtrec.serving.app
(/root/reference/rtrec/serving/app.py:35-93).

Same routes, request/response payloads, X-Token check and status codes: GET / (liveness),
POST /fit (a list of interactions -> SLIM.fit, i.e. ingest + refit of the touched item columns on
the GPU), POST /recommend (one user's top-k).  Differences, all forced by the device:
  * `ModelGate` serialises every model call with one lock -- the engine owns one HIP stream and is
    not re-entrant (SURVEY.md section 8b "Threading"; the reference calls its model unlocked);
  * handlers are plain functions, so FastAPI runs them in its worker pool and a long /fit does
    not stall the event loop;
  * POST /recommend_batch is an addition for callers that can batch (one kernel launch per request
    instead of one per user): {"users": [...]} -> {"users": [...], "recommendations": [[...], ...]};
  * concurrent POST /recommend calls are coalesced (`RecommendCoalescer`): requests that arrive within a bounded
    wait (RTREC_AMD_COALESCE_MS, default 1 ms; 0 = only what queued up behind the model lock) share ONE
    recommend_batch launch per (top_k, filter_interacted) group; each caller gets exactly what its own
    model.recommend call would have returned, and a failure of the shared call falls back to one call per request.
"""
from __future__ import annotations

import logging
import os
import threading
import time
from typing import Any, Callable, Dict, List, Optional, Tuple

from fastapi import APIRouter, FastAPI, Header, HTTPException
from fastapi.middleware.cors import CORSMiddleware
from pydantic import BaseModel
from .. import settings

SECRET_TOKEN = os.getenv("X_TOKEN", "fake_secret_token")
log = logging.getLogger("rtrec_amd.serving")


class Interaction(BaseModel):
    user: Any
    item: Any
    timestamp: float
    rating: float


class _TopKOptions(BaseModel):
    top_k: int = 10
    filter_interacted: bool = True


class RecommendationRequest(_TopKOptions):
    user: Any


class BatchRecommendationRequest(_TopKOptions):
    users: List[Any]


class RecommendationResponse(BaseModel):
    user: Any
    recommendations: List[Any]


class BatchRecommendationResponse(BaseModel):
    users: List[Any]
    recommendations: List[List[Any]]


class ModelGate:
    """The model plus the lock that serialises GPU work; failures become the reference's 500s."""

    def __init__(self, model: Any, coalesce_ms: Optional[float] = None):
        self.model = model
        self._lock = threading.Lock()
        ms = float(settings.raw("RTREC_AMD_COALESCE_MS", "1")) if coalesce_ms is None else float(coalesce_ms)
        self.coalescer = RecommendCoalescer(self, max_wait_s=ms * 1e-3) if ms >= 0 else None

    def recommend(self, user: Any, top_k: int, filter_interacted: bool) -> Any:
        """One /recommend request: coalesced with its concurrent neighbours (RTREC_AMD_COALESCE_MS < 0 disables it)."""
        if self.coalescer is None:
            return self.call("Recommendation", lambda m: m.recommend(user=user, top_k=top_k, filter_interacted=filter_interacted))
        try:
            return self.coalescer.submit(user, top_k, filter_interacted)
        except Exception as exc:
            log.error("Recommendation failed: %s", exc)
            raise HTTPException(status_code=500, detail="Recommendation failed")

    def call(self, what: str, fn: Callable[[Any], Any]) -> Any:
        try:
            with self._lock:
                return fn(self.model)
        except Exception as exc:  # same catch-all as the reference handlers
            log.error("%s failed: %s", what, exc)
            raise HTTPException(status_code=500, detail=f"{what} failed")


class _Pending:
    __slots__ = ("user", "top_k", "filter_interacted", "event", "done", "result", "error")

    def __init__(self, user: Any, top_k: int, filter_interacted: bool):
        self.user, self.top_k, self.filter_interacted = user, top_k, filter_interacted
        self.event = threading.Event()
        self.done = False
        self.result: Any = None
        self.error: Optional[BaseException] = None


class RecommendCoalescer:
    """Leader/follower batching of single-user requests (no extra thread): the first caller to find nobody leading
    becomes the leader, waits at most `max_wait_s` for company (or until `max_batch` requests are queued), takes the
    model lock, drains the queue and answers the whole batch with one recommend_batch launch per option group; if
    requests are still queued afterwards it hands the lead to the oldest of them."""

    def __init__(self, gate: "ModelGate", max_wait_s: float = 0.001, max_batch: int = 256):
        self.gate, self.max_wait_s, self.max_batch = gate, max(0.0, float(max_wait_s)), max(1, int(max_batch))
        self._mu = threading.Lock()
        self._queue: List[_Pending] = []
        self._leading = False
        self._full = threading.Event()
        self._last_other = -1e9     # when a request last arrived while another was in flight
        self.rounds = 0             # recommend launches made on behalf of /recommend
        self.requests = 0           # /recommend calls answered

    def submit(self, user: Any, top_k: int, filter_interacted: bool) -> Any:
        p = _Pending(user, top_k, filter_interacted)
        with self._mu:
            self._queue.append(p)
            if self._leading:
                self._last_other = time.monotonic()      # a request arrived while another was being answered: callers overlap
            lead = not self._leading
            if lead:
                self._leading = True
            elif len(self._queue) >= self.max_batch:
                self._full.set()
        while True:
            if lead:
                self._round()
            p.event.wait()
            if p.done:
                break
            p.event.clear()         # woken without an answer: promoted to leader
            lead = True
        if p.error is not None:
            raise p.error
        return p.result

    def _round(self) -> None:
        # the bounded wait collects concurrent callers; a request that is alone -- nothing else queued, no other request seen
        # within the last window -- goes straight through (an isolated POST /recommend paid the whole wait)
        now = time.monotonic()
        with self._mu:
            crowded = len(self._queue) > 1 or (now - self._last_other) < 4.0 * self.max_wait_s
        if self.max_wait_s > 0.0 and crowded:
            self._full.wait(self.max_wait_s)
        batch: List[_Pending] = []
        try:
            with self.gate._lock:                       # requests keep queueing while a /fit holds the model
                with self._mu:
                    batch, self._queue = self._queue[:self.max_batch], self._queue[self.max_batch:]
                    self._full.clear()
                self._answer(batch)
        except BaseException as exc:                    # never leave a follower waiting
            for p in batch:
                if not p.done and p.error is None and p.result is None:
                    p.error = exc
            raise
        finally:
            with self._mu:
                self.requests += len(batch)
                if self._queue:
                    self._queue[0].event.set()          # done is False: that caller leads the next round
                else:
                    self._leading = False
            for p in batch:
                p.done = True
                p.event.set()

    def _answer(self, batch: List[_Pending]) -> None:
        m = self.gate.model
        groups: Dict[Tuple[int, bool], List[_Pending]] = {}
        for p in batch:
            groups.setdefault((p.top_k, p.filter_interacted), []).append(p)
        for (top_k, filt), ps in groups.items():
            # users the model does not know take the reference's cold-start branch of recommend() one by one
            known = [p for p in ps if self._known(m, p.user)]
            if len(known) > 1:
                try:
                    rows = m.recommend_batch([p.user for p in known], top_k=top_k, filter_interacted=filt)
                    self.rounds += 1
                    for p, row in zip(known, rows):
                        p.result = row
                except Exception as exc:
                    log.warning("coalesced recommend failed (%s): answering %d requests one by one", exc, len(known))
                    known = []
            else:
                known = []
            for p in ps:
                if p.result is None and p.error is None and p not in known:
                    try:
                        p.result = m.recommend(user=p.user, top_k=top_k, filter_interacted=filt)
                        self.rounds += 1
                    except Exception as exc:
                        p.error = exc

    @staticmethod
    def _known(m: Any, user: Any) -> bool:
        try:
            return m._known_user_id(user) is not None
        except Exception:
            return False


def _authorise(x_token: str) -> None:
    if x_token != SECRET_TOKEN:
        raise HTTPException(status_code=400, detail="Invalid X-Token header")


def build_router(gate: ModelGate) -> APIRouter:
    api = APIRouter()

    @api.get("/")
    def alive():
        return {"message": "Recommender System API is running"}

    @api.post("/fit")
    def fit(interactions: List[Interaction], x_token: str = Header()):
        _authorise(x_token)
        batch = [(row.user, row.item, row.timestamp, row.rating) for row in interactions]
        gate.call("Training", lambda m: m.fit(batch, progress_bar=False))
        return {"message": "Training successful"}

    @api.post("/recommend", response_model=RecommendationResponse)
    def recommend(request: RecommendationRequest, x_token: str = Header()):
        _authorise(x_token)
        items = gate.recommend(request.user, request.top_k, request.filter_interacted)
        return RecommendationResponse(user=request.user, recommendations=items)

    @api.post("/recommend_batch", response_model=BatchRecommendationResponse)
    def recommend_batch(request: BatchRecommendationRequest, x_token: str = Header()):
        _authorise(x_token)
        lists = gate.call("Recommendation", lambda m: m.recommend_batch(
            request.users, top_k=request.top_k, filter_interacted=request.filter_interacted))
        return BatchRecommendationResponse(users=request.users, recommendations=lists)

    return api


def create_app(model_factory: Optional[Callable[[], Any]] = None) -> FastAPI:
    """App factory.  `model_factory` builds the model; the default is the reference's
    SLIM(min_value=-5, max_value=10, decay_in_days=365) (app.py:49).  Tests pass a factory whose
    engine runs on the CPU oracle backend."""
    if model_factory is None:
        from ..models.slim import SLIM

        def model_factory():
            return SLIM(min_value=-5, max_value=10, decay_in_days=365)
    app = FastAPI()
    app.add_middleware(CORSMiddleware, allow_origins=["*"], allow_credentials=True, allow_methods=["*"],
                       allow_headers=["*"])
    app.include_router(build_router(ModelGate(model_factory())))
    return app


if __name__ == "__main__":
    import uvicorn
    uvicorn.run(create_app(), host="0.0.0.0", port=8000)
