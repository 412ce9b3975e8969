"""Materialise a LAYER RANGE of a Llama checkpoint on one device (one pipeline rank = one process = one GPU).

The reference loads every checkpoint whole and lets accelerate spread it (`device_map="auto"`, reference generate.py:54-67):
one process drives all GPUs through forward hooks that copy activations from device to device.  The MI355X-native partition
is one process per GPU with point-to-point hand-offs (layerskip_amd/pipeline.py), so a rank needs exactly its decoder layers
`[a, b)` plus the small shared tensors (embedding, final norm, lm_head: rank 0 drafts with its own head copy, the last rank runs
the verify head) -- and must never pull the other 7/8 of a 140 GB checkpoint through its HBM.

`--model` may be a directory or a hub id (`facebook/layerskip-llama2-7B`, the reference's usage, generate.py:59-64): a name that is
not a directory is resolved to its snapshot directory through `huggingface_hub.snapshot_download` (config, tokenizer and safetensors
files only; with HF_HUB_OFFLINE=1 that is a pure cache lookup).

`load_layer_range` reads a `save_pretrained` / hub-layout directory (`config.json` + `model.safetensors` or the sharded
`model-0000x-of-0000y.safetensors` with `model.safetensors.index.json`), opens only the shard files that hold tensors this rank
owns and copies those tensors straight to the rank's device.  Decoder layers outside the range stay on the meta device (no
storage at all); `HipEngine(model, layer_range=(a, b))` only touches the owned ones.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, Optional, Sequence, Tuple

import torch
import transformers

INDEX_NAME = "model.safetensors.index.json"
SINGLE_NAME = "model.safetensors"


_META_PATTERNS = ["*.json", "tokenizer*", "*.model"]      # config, generation config, the safetensors index, the tokenizer: kilobytes
_RESOLVED: Dict[str, str] = {}                             # name -> snapshot directory, once per process


def _hub_download(name: str, patterns) -> str:
    """`snapshot_download` of the files matching `patterns`.  What the hub says is kept apart: a name that is nowhere (no repository,
    or offline without a cached snapshot) is a FileNotFoundError; a network, authentication or disk failure stays what it is."""
    from huggingface_hub import snapshot_download
    try:
        return snapshot_download(name, allow_patterns=list(patterns))
    except Exception as exc:       # noqa: BLE001 -- classified below
        import huggingface_hub.errors as hub_errors
        missing = tuple(getattr(hub_errors, n) for n in ("RepositoryNotFoundError", "LocalEntryNotFoundError", "RevisionNotFoundError",
                                                           "HFValidationError") if hasattr(hub_errors, n))
        if isinstance(exc, missing + (FileNotFoundError,)):
            raise FileNotFoundError(f"{name!r} is neither a checkpoint directory nor a hub id that could be resolved to a snapshot directory "
                                    f"({type(exc).__name__}: {exc}).  Pass a `save_pretrained` directory, or populate the hub cache "
                                    f"(`huggingface-cli download {name}`); only safetensors checkpoints are read") from exc
        raise RuntimeError(f"resolving the hub id {name!r} failed ({type(exc).__name__}: {exc}): a network, authentication or cache-disk "
                           f"problem, not a missing checkpoint") from exc


def _local_barrier_leader() -> Optional[bool]:
    """Under torchrun with an initialised process group: is this the rank that downloads for its node?  None outside one."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return None
    return int(os.environ.get("LOCAL_RANK", str(dist.get_rank()))) == 0


def resolve_checkpoint_dir(path: str) -> str:
    """A checkpoint DIRECTORY for what the reference hands to `from_pretrained` (generate.py:59-64): a local directory as it is, a hub id
    (or anything else `from_pretrained` would accept by name) as its snapshot directory in the hub cache.  Resolved ONCE per process
    (the drivers ask three times: partition, tokenizer, weights); only the metadata -- config, safetensors index, tokenizer -- is fetched
    here, the weight shards are `load_layer_range`'s business, restricted to the files that hold the rank's tensors (a 70B checkpoint is
    140 GB; a pipeline rank reads an eighth of it).  In a multi-process run the local rank 0 of a node fetches, the others wait at a
    barrier and then find the files in the shared cache -- one etag round per node instead of one per rank."""
    if os.path.isdir(path):
        return path
    if path in _RESOLVED:
        return _RESOLVED[path]
    leader = _local_barrier_leader()
    err = None
    out = None
    if leader is None or leader:
        try:
            out = _hub_download(path, _META_PATTERNS)
        except Exception as exc:       # noqa: BLE001 -- re-raised below, AFTER the barrier the other ranks wait at
            err = exc
    if leader is not None:
        import torch.distributed as dist
        dist.barrier()
        if not leader:
            out = _hub_download(path, _META_PATTERNS)       # now a cache hit (or the same error the leader saw)
    if err is not None:
        raise err
    _RESOLVED[path] = out
    return out


def _fetch_shards(name: str, directory: str, files: Iterable[str]) -> None:
    """The shard files of a hub snapshot this rank reads, if they are not in the snapshot directory yet (every rank fetches ITS files:
    mostly disjoint sets, so the ranks of a node share the download instead of repeating it)."""
    need = sorted(f for f in set(files) if not os.path.exists(os.path.join(directory, f)))
    if need:
        _hub_download(name, need)


def _weight_map(path: str, hub_name: Optional[str] = None) -> Dict[str, str]:
    """tensor name -> file name (relative to `path`).  hub_name: `path` is the snapshot directory of this hub id (an unsharded
    checkpoint's one weight file is fetched here: there is no index to learn the file names from)."""
    index = os.path.join(path, INDEX_NAME)
    if os.path.exists(index):
        with open(index) as f:
            return dict(json.load(f)["weight_map"])
    single = os.path.join(path, SINGLE_NAME)
    if hub_name is not None and not os.path.exists(single):
        try:
            _hub_download(hub_name, [SINGLE_NAME])
        except FileNotFoundError:
            pass                   # reported below, with the directory's name
    if os.path.exists(single):
        from safetensors import safe_open
        with safe_open(single, framework="pt", device="cpu") as f:
            return {k: SINGLE_NAME for k in f.keys()}
    raise FileNotFoundError(f"{path}: neither {INDEX_NAME} nor {SINGLE_NAME} (only safetensors checkpoints are read; the reference "
                            f"passes use_safetensors=True as well, generate.py:61)")


def owned_parameter_names(names: Iterable[str], layer_range: Optional[Sequence[int]], embed: bool = True, head: bool = True) -> list:
    """The parameters a rank materialises: its decoder layers, the (tiny) final norm, and -- where the rank uses them -- the embedding
    (rank 0: the input rows and the drafted tokens) and the lm_head (rank 0: the draft head; the last rank: the verify head).  A middle
    rank of a pipeline needs neither: 2 x 2.1 GB at llama3-70B that never have to cross its HBM."""
    out = []
    for name in names:
        if layer_range is not None and name.startswith("model.layers."):
            if not (layer_range[0] <= int(name.split(".")[2]) < layer_range[1]):
                continue
        if name == "model.embed_tokens.weight" and not embed:
            continue
        if name == "lm_head.weight" and not head:
            continue
        out.append(name)
    return out


def _assign(model: torch.nn.Module, dotted: str, value: torch.Tensor) -> None:
    mod = model
    parts = dotted.split(".")
    for p in parts[:-1]:
        mod = getattr(mod, p)
    setattr(mod, parts[-1], torch.nn.Parameter(value, requires_grad=False))


@torch.no_grad()
def load_layer_range(path: str, layer_range: Optional[Sequence[int]] = None, device: str | torch.device = "cuda:0",
                     dtype: torch.dtype = torch.bfloat16, embed: bool = True, head: bool = True) -> transformers.LlamaForCausalLM:
    """`LlamaForCausalLM` with decoder layers `[a, b)` (all of them for `layer_range=None`), the final norm and -- unless the caller
    says this rank uses neither (`embed=False` / `head=False`: a middle rank of a pipeline) -- the embedding and the lm_head on `device`
    in `dtype`; everything else on the meta device.  `path`: a directory or a hub id.  `model.loaded_layer_range` = (a, b)."""
    device = torch.device(device)
    hub_name = None if os.path.isdir(path) else path
    path = resolve_checkpoint_dir(path)
    config = transformers.AutoConfig.from_pretrained(path)
    if getattr(config, "model_type", "llama") != "llama":
        raise ValueError(f"{path}: model_type {config.model_type!r}; the engine implements the Llama decoder (SURVEY.md 8a)")
    n_layers = config.num_hidden_layers
    rng: Tuple[int, int] = (0, n_layers) if layer_range is None else (int(layer_range[0]), int(layer_range[1]))
    if not (0 <= rng[0] < rng[1] <= n_layers):
        raise ValueError(f"layer_range {rng} outside [0, {n_layers}]")
    with torch.device("meta"):
        model = transformers.LlamaForCausalLM(config)
    model.eval()
    wmap = _weight_map(path, hub_name)
    tied = bool(getattr(config, "tie_word_embeddings", False))
    # a tied checkpoint's head IS its embedding: a rank that runs a head needs that tensor
    wanted = owned_parameter_names([n for n, _ in model.named_parameters()], rng, embed=embed or (head and tied), head=head)
    by_file: Dict[str, list] = {}
    for name in wanted:
        if name == "lm_head.weight" and name not in wmap:
            if tied:
                continue            # tied checkpoints store the embedding only
            raise KeyError(f"{path}: lm_head.weight is missing and the config does not tie it to the embedding")
        if name not in wmap:
            raise KeyError(f"{path}: tensor {name} not in the checkpoint")
        by_file.setdefault(wmap[name], []).append(name)
    if hub_name is not None:
        _fetch_shards(hub_name, path, by_file)          # only the files that hold this rank's tensors
    from safetensors import safe_open
    for fname, names in sorted(by_file.items()):
        # only the shard files that hold owned tensors are opened; tensors go file -> device without a host-side model copy
        with safe_open(os.path.join(path, fname), framework="pt", device=str(device)) as f:
            for name in names:
                _assign(model, name, f.get_tensor(name).to(dtype).contiguous())
    if head and (tied or model.lm_head.weight.device.type == "meta"):
        model.lm_head.weight = model.model.embed_tokens.weight
    # non-persistent buffers (rotary inv_freq) were created on the meta device: rebuild them
    model.model.rotary_emb = type(model.model.rotary_emb)(config).to(device)
    model.requires_grad_(False)
    model.loaded_layer_range = rng
    return model
