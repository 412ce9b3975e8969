"""MI355X-native self-speculative decoding engine behind LayerSkip's GenerationStrategy surface.

Importing the package needs neither a GPU nor the built extension; constructing an engine does
(there is no CPU fallback -- see ``layerskip_amd._lib.load``).
"""
from .strategy_api import (GenerationConfig, GenerationResult, GenerationStrategy,  # noqa: F401
                           GenerationStrategyResult, TokenGenerator)

__all__ = ["GenerationConfig", "GenerationResult", "GenerationStrategy", "GenerationStrategyResult",
           "TokenGenerator", "HipEngine", "get_engine", "HipSelfSpeculativeGenerationStrategy",
           "HipAutoRegressiveGenerationStrategy"]


def __getattr__(name):
    if name in ("HipEngine", "get_engine"):
        from . import engine
        return getattr(engine, name)
    if name in ("HipSelfSpeculativeGenerationStrategy", "HipAutoRegressiveGenerationStrategy", "STRATEGIES"):
        from . import hip_strategies
        return getattr(hip_strategies, name)
    raise AttributeError(name)
