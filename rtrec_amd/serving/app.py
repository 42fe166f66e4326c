"""HTTP shell around the GPU SLIM model: the drop-in for rtrec.serving.app
(/root/reference/rtrec/serving/app.py:35-93).

Same routes, request/response models, X-Token check and status codes: GET / (liveness),
POST /fit (a list of interactions -> SLIM.fit, i.e. ingest + refit of the touched item columns on
the GPU), POST /recommend (one user's top-k).  Two differences, both forced by the device:
  * one lock serialises model calls -- the engine owns one HIP stream and is not re-entrant
    (SURVEY.md section 8b "Threading"; the reference calls its model with no lock, app.py:57-91);
  * the handlers are plain `def`, so FastAPI runs them in its worker pool and a long /fit does
    not stall the event loop (the reference's `async def` handlers block it).
POST /recommend_batch is an addition for callers that can batch (one kernel launch per request
instead of one per user); it returns {"users": [...], "recommendations": [[...], ...]}.
"""
from __future__ import annotations

import logging
import os
import threading
from typing import Any, Callable, List, Optional

from fastapi import FastAPI, Header, HTTPException
from fastapi.middleware.cors import CORSMiddleware
from pydantic import BaseModel

DEFAULT_SECRET_TOKEN = "fake_secret_token"
SECRET_TOKEN = os.getenv("X_TOKEN", DEFAULT_SECRET_TOKEN)


class Interaction(BaseModel):
    user: Any
    item: Any
    timestamp: float
    rating: float


class RecommendationRequest(BaseModel):
    user: Any
    top_k: int = 10
    filter_interacted: bool = True


class RecommendationResponse(BaseModel):
    user: Any
    recommendations: List[Any]


class BatchRecommendationRequest(BaseModel):
    users: List[Any]
    top_k: int = 10
    filter_interacted: bool = True


class BatchRecommendationResponse(BaseModel):
    users: List[Any]
    recommendations: List[List[Any]]


def create_app(model_factory: Optional[Callable[[], Any]] = None) -> FastAPI:
    """App factory.  `model_factory` builds the model (default: the reference's
    SLIM(min_value=-5, max_value=10, decay_in_days=365), app.py:49); tests pass a factory whose
    engine runs on the CPU oracle backend."""
    app = FastAPI()
    app.add_middleware(CORSMiddleware, allow_origins=["*"], allow_credentials=True, allow_methods=["*"],
                       allow_headers=["*"])
    if model_factory is None:
        from ..models.slim import SLIM
        recommender = SLIM(min_value=-5, max_value=10, decay_in_days=365)
    else:
        recommender = model_factory()
    lock = threading.Lock()

    def check(x_token: str) -> None:
        if x_token != SECRET_TOKEN:
            raise HTTPException(status_code=400, detail="Invalid X-Token header")

    @app.get("/")
    def read_root():
        return {"message": "Recommender System API is running"}

    @app.post("/fit")
    def fit(interactions: List[Interaction], x_token: str = Header()):
        check(x_token)
        try:
            rows = [(i.user, i.item, i.timestamp, i.rating) for i in interactions]
            with lock:
                recommender.fit(rows, progress_bar=False)
            return {"message": "Training successful"}
        except Exception as e:
            logging.error(f"Training failed: {e}")
            raise HTTPException(status_code=500, detail="Training failed")

    @app.post("/recommend", response_model=RecommendationResponse)
    def recommend(request: RecommendationRequest, x_token: str = Header()):
        check(x_token)
        try:
            with lock:
                recs = recommender.recommend(user=request.user, top_k=request.top_k,
                                             filter_interacted=request.filter_interacted)
            return {"user": request.user, "recommendations": recs}
        except Exception as e:
            logging.error(f"Recommendation failed: {e}")
            raise HTTPException(status_code=500, detail="Recommendation failed")

    @app.post("/recommend_batch", response_model=BatchRecommendationResponse)
    def recommend_batch(request: BatchRecommendationRequest, x_token: str = Header()):
        check(x_token)
        try:
            with lock:
                recs = recommender.recommend_batch(request.users, top_k=request.top_k,
                                                   filter_interacted=request.filter_interacted)
            return {"users": request.users, "recommendations": recs}
        except Exception as e:
            logging.error(f"Recommendation failed: {e}")
            raise HTTPException(status_code=500, detail="Recommendation failed")

    return app


if __name__ == "__main__":
    import uvicorn
    uvicorn.run(create_app(), host="0.0.0.0", port=8000)
