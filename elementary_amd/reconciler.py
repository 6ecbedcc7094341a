"""Instruction-batch frontend: hash-consed node tree, reconciler and Renderer.

Mirror of the reference's JS frontend for the one thing the hot path needs from it:
the *instruction wire format* consumed by ``Runtime::applyInstructions``
(runtime/elem/Runtime.h:170-218).  Batches produced here are byte-compatible with what
``@elemaudio/core`` emits, including the int32 node hashes, so the reference's own
reconciler snapshots (js/packages/core/__tests__/__snapshots__/core.test.js.snap) serve as
known-answer tests (tests/test_reconciler.py).

Reference files restated (behaviour, not code):
  * FNV-1a node hashing ............ js/packages/core/src/HashUtils.res:8-44
  * NodeRepr.create ................ js/packages/core/src/NodeRepr.res:34-54
  * mount / visit / render ......... js/packages/core/src/Reconciler.res:45-100
  * Delegate batching + Renderer ... js/packages/core/index.ts:53-226
  * updateNodeProps ................ js/packages/core/src/Hash.ts:4-31
"""
from __future__ import annotations

import json
import math
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

CREATE_NODE, APPEND_CHILD, SET_PROPERTY, ACTIVATE_ROOTS, COMMIT_UPDATES = 0, 2, 3, 4, 5

_M32 = 0xFFFFFFFF


def _to_i32(x: int) -> int:
    x &= _M32
    return x - (1 << 32) if x & 0x80000000 else x


def mix_number(seed: int, n: int) -> int:
    """HashUtils.res:22-24 — (seed ^ n) * 0x01000193 on wrapped signed 32-bit ints."""
    return _to_i32(((seed ^ n) & _M32) * 0x01000193)


def hash_string(seed: int, s: str) -> int:
    """HashUtils.res:27-35. The ReScript loop bound is inclusive (``0 to length``), so one
    extra step mixes charCodeAt(length) = NaN -> 0."""
    r = seed
    units = s.encode("utf-16-le")
    for i in range(0, len(units), 2):
        r = mix_number(r, units[i] | (units[i + 1] << 8))
    return mix_number(r, 0)


def js_number(x: float) -> str:
    """ECMAScript Number::toString(10) — what JSON.stringify prints for a number."""
    x = float(x)
    if x != x or x in (math.inf, -math.inf):
        return "null"  # JSON.stringify(NaN/Infinity)
    if x == 0:
        return "0"
    sign = "-" if x < 0 else ""
    r = repr(abs(x))
    if "e" in r:
        mant, exp = r.split("e")
        e = int(exp)
    else:
        mant, e = r, 0
    if "." in mant:
        ip, fp = mant.split(".")
    else:
        ip, fp = mant, ""
    digits = (ip + fp).lstrip("0")
    # decimal point position relative to the start of `digits`
    n = len(ip) + e - (len(ip + fp) - len((ip + fp).lstrip("0")))
    digits = digits.rstrip("0") or "0"
    k = len(digits)
    if k <= n <= 21:
        return sign + digits + "0" * (n - k)
    if 0 < n <= 21:
        return sign + digits[:n] + "." + digits[n:]
    if -6 < n <= 0:
        return sign + "0." + "0" * (-n) + digits
    ex = n - 1
    es = ("+" if ex >= 0 else "-") + str(abs(ex))
    if k == 1:
        return sign + digits + "e" + es
    return sign + digits[0] + "." + digits[1:] + "e" + es


def js_stringify(v: Any) -> str:
    """JSON.stringify for the prop values the frontend uses (insertion-ordered keys)."""
    if v is None:
        return "null"
    if v is True:
        return "true"
    if v is False:
        return "false"
    if isinstance(v, (int, float)):
        return js_number(v)
    if isinstance(v, str):
        return json.dumps(v, ensure_ascii=False)
    if isinstance(v, (list, tuple)):
        return "[" + ",".join(js_stringify(x) for x in v) + "]"
    if isinstance(v, dict):
        return "{" + ",".join(json.dumps(str(k), ensure_ascii=False) + ":" + js_stringify(x) for k, x in v.items()) + "}"
    raise TypeError(f"cannot stringify {type(v)}")


class NodeRepr:
    """Hash-consed node (NodeRepr.res:11-54)."""

    __slots__ = ("kind", "props", "children", "hash", "output_channel")

    def __init__(self, kind: str, props: Dict[str, Any], children: Sequence["NodeRepr"], output_channel: int = 0, _hash: Optional[int] = None):
        self.kind = kind
        self.props = props
        self.children = tuple(children)
        self.output_channel = output_channel
        self.hash = _hash if _hash is not None else hash_node(kind, props, [mix_number(c.hash, c.output_channel) for c in self.children])

    def with_output_channel(self, ch: int) -> "NodeRepr":
        return NodeRepr(self.kind, self.props, self.children, ch, self.hash)


def hash_node(kind: str, props: Dict[str, Any], children: Iterable[int]) -> int:
    """HashUtils.res:37-45."""
    r = hash_string(_to_i32(0x811C9DC5), kind)
    key = props.get("key")
    if isinstance(key, str):
        r = hash_string(r, key)
    else:
        r = hash_string(r, js_stringify(props))
    for c in children:
        r = mix_number(r, c)
    return r & 0x7FFFFFFF


ElemNode = Any  # NodeRepr | number


def resolve(n: ElemNode) -> NodeRepr:
    """nodeUtils.ts:11-17 — numbers become const nodes."""
    if isinstance(n, NodeRepr):
        return n
    if isinstance(n, bool) or not isinstance(n, (int, float)):
        raise TypeError(f"expecting a node or a number, got {type(n)}")
    return NodeRepr("const", {"value": n}, [])


def create_node(kind: str, props: Optional[Dict[str, Any]], children: Sequence[ElemNode]) -> NodeRepr:
    return NodeRepr(kind, dict(props or {}), [resolve(c) for c in children])


def unpack(node: NodeRepr, num_channels: int) -> List[NodeRepr]:
    return [node.with_output_channel(i) for i in range(num_channels)]


def _shallow_equal(a: Any, b: Any) -> bool:
    if a is b:
        return True
    if type(a) is not type(b) and not (isinstance(a, (int, float)) and isinstance(b, (int, float))):
        return False
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(x is y or (not isinstance(x, (list, dict)) and x == y) for x, y in zip(a, b))
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(a[k] is b[k] or (not isinstance(a[k], (list, dict)) and a[k] == b[k]) for k in a)
    return a == b


class Delegate:
    """Batching render delegate (index.ts:53-131)."""

    def __init__(self) -> None:
        self.node_map: Dict[int, Dict[str, Any]] = {}
        self.current_active_roots: set = set()
        self.clear()

    def clear(self) -> None:
        self.nodes_added = self.edges_added = self.props_written = 0
        self._create: List[list] = []
        self._append: List[list] = []
        self._props: List[list] = []
        self._activate: List[list] = []
        self._commit: List[list] = []

    def create_node(self, h: int, kind: str) -> None:
        self.nodes_added += 1
        self._create.append([CREATE_NODE, h, kind])

    def append_child(self, parent: int, child: int, ch: int) -> None:
        self.edges_added += 1
        self._append.append([APPEND_CHILD, parent, child, ch])

    def set_property(self, h: int, key: str, value: Any) -> None:
        self.props_written += 1
        self._props.append([SET_PROPERTY, h, key, value])

    def activate_roots(self, roots: List[int]) -> None:
        already = len(roots) == len(self.current_active_roots) and all(r in self.current_active_roots for r in roots)
        if not already:
            self._activate.append([ACTIVATE_ROOTS, list(roots)])
            self.current_active_roots = set(roots)

    def commit_updates(self) -> None:
        self._commit.append([COMMIT_UPDATES])

    def packed(self) -> List[list]:
        return self._create + self._append + self._props + self._activate + self._commit


def update_node_props(delegate: Delegate, h: int, prev: Dict[str, Any], nxt: Dict[str, Any]) -> None:
    """Hash.ts:4-31."""
    for key, value in nxt.items():
        if key not in prev or not _shallow_equal(prev[key], value):
            delegate.set_property(h, key, value)
            prev[key] = value


def render_with_delegate(delegate: Delegate, graphs: Sequence[NodeRepr], fade_in_ms: float = 20, fade_out_ms: float = 20) -> List[int]:
    """Reconciler.res:45-100: pre-order visit with a visited set, mount new hashes only."""
    roots = [NodeRepr("root", {"channel": i, "fadeInMs": fade_in_ms, "fadeOutMs": fade_out_ms}, [g]) for i, g in enumerate(graphs)]
    visited: set = set()
    stack: List[NodeRepr] = list(reversed(roots))
    node_map = delegate.node_map
    while stack:
        n = stack.pop()
        if n.hash in visited:
            continue
        visited.add(n.hash)
        existing = node_map.get(n.hash)
        if existing is None:
            delegate.create_node(n.hash, n.kind)
            shadow: Dict[str, Any] = {}
            update_node_props(delegate, n.hash, shadow, n.props)
            for c in n.children:
                delegate.append_child(n.hash, c.hash, c.output_channel)
            node_map[n.hash] = {"kind": n.kind, "props": shadow}
        else:
            update_node_props(delegate, n.hash, existing["props"], n.props)
        stack.extend(reversed(n.children))
    hashes = [r.hash for r in roots]
    delegate.activate_roots(hashes)
    delegate.commit_updates()
    return hashes


class Renderer:
    """index.ts:142-236 — ``render(*roots)`` sends one packed batch through ``send``."""

    def __init__(self, send: Callable[[List[list]], Any]):
        self._delegate = Delegate()
        self._send = send
        self._next_ref = 0

    def create_ref(self, kind: str, props: Dict[str, Any], children: Sequence[ElemNode]) -> Tuple[NodeRepr, Callable[[Dict[str, Any]], Any]]:
        key = f"__refKey:{self._next_ref}"
        self._next_ref += 1
        p = {"key": key}
        p.update(props)
        node = create_node(kind, p, children)

        def setter(new_props: Dict[str, Any]) -> Any:
            if node.hash not in self._delegate.node_map:
                raise RuntimeError("Cannot update a ref that has not been mounted; make sure you render your node first")
            self._delegate.clear()
            update_node_props(self._delegate, node.hash, self._delegate.node_map[node.hash]["props"], new_props)
            self._delegate.commit_updates()
            return self._send(self._delegate.packed())

        return node, setter

    def render(self, *roots: ElemNode, root_fade_in_ms: float = 20, root_fade_out_ms: float = 20) -> Dict[str, Any]:
        self._delegate.clear()
        render_with_delegate(self._delegate, [resolve(r) for r in roots], root_fade_in_ms, root_fade_out_ms)
        batch = self._delegate.packed()
        result = self._send(batch)
        if result != 0:
            # the engine stopped somewhere inside the batch: force ACTIVATE_ROOTS (and with it a rebuild) into the next render
            self._delegate.current_active_roots = set()
        return {
            "result": result,
            "nodesAdded": self._delegate.nodes_added,
            "edgesAdded": self._delegate.edges_added,
            "propsWritten": self._delegate.props_written,
            "batch": batch,
        }

    def prune(self, node_ids: Iterable[int]) -> None:
        for n in node_ids:
            self._delegate.node_map.pop(n, None)


def batch_to_json(batch: List[list]) -> str:
    """Serialise a batch the way the cli host receives it (cli/Benchmark.cpp:40-43)."""
    return json.dumps(batch, separators=(",", ":"), ensure_ascii=False)
