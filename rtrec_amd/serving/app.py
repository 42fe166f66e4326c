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
    instead of one per user): {"users": [...]} -> {"users": [...], "recommendations": [[...], ...]}.
"""
from __future__ import annotations

import logging
import os
import threading
from typing import Any, Callable, List, Optional

from fastapi import APIRouter, FastAPI, Header, HTTPException
from fastapi.middleware.cors import CORSMiddleware
from pydantic import BaseModel

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

    def __init__(self, model: Any):
        self.model = model
        self._lock = threading.Lock()

    def call(self, what: str, fn: Callable[[Any], Any]) -> Any:
        try:
            with self._lock:
                return fn(self.model)
        except Exception as exc:  # same catch-all as the reference handlers
            log.error("%s failed: %s", what, exc)
            raise HTTPException(status_code=500, detail=f"{what} failed")


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
        items = gate.call("Recommendation", lambda m: m.recommend(
            user=request.user, top_k=request.top_k, filter_interacted=request.filter_interacted))
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
