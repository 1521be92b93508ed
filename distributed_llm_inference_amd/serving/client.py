"""Client library and interactive chat for the orchestrator API.

Same surface as the reference's `DistributedLLMClient` (Test.py:8-144): `check_health()`,
`check_workers()`, `generate(prompt, max_tokens=20)`, `interactive_chat()` with the `quit` /
`exit` / `q`, `workers` and `health` commands, and the three-way start menu (Test.py:147-187).
"""
from __future__ import annotations

import sys
from typing import Dict, Optional


class DistributedLLMClient:
    def __init__(self, api_url: str, timeout: float = 200.0, verbose: bool = True, check: bool = True):
        import requests
        self._rq = requests
        self.api_url = api_url.rstrip("/")
        self.headers = {"ngrok-skip-browser-warning": "true", "User-Agent": "DistributedLLMClient/1.0"}
        self.timeout = timeout            # Test.py:71
        self.verbose = verbose
        self._say(f"Connected to: {self.api_url}")
        if check:
            self.check_health()

    def _say(self, *a, **kw):
        if self.verbose:
            print(*a, **kw)

    def check_health(self) -> bool:
        try:
            r = self._rq.get(f"{self.api_url}/health", headers=self.headers, timeout=5)
            if r.status_code == 200:
                d = r.json()
                self._say(f"API Status: {d.get('status', 'unknown')}")
                self._say(f"   Role: {d.get('role', 'unknown')}")
                return True
            return False
        except Exception as e:
            self._say(f"Cannot connect: {e}")
            return False

    def check_workers(self) -> Optional[Dict[str, str]]:
        try:
            r = self._rq.get(f"{self.api_url}/workers", headers=self.headers, timeout=5)
            if r.status_code == 200:
                d = r.json()
                self._say("\nWorker Status:")
                for w, st in d.items():
                    self._say(f"   [{'ok' if st == 'online' else '!!'}] {w}: {st}")
                return d
            return None
        except Exception as e:
            self._say(f"Error: {e}")
            return None

    def generate(self, prompt: str, max_tokens: int = 20, temperature: Optional[float] = None) -> Optional[Dict]:
        self._say("\n" + "=" * 60)
        self._say(f"Prompt: {prompt}")
        self._say("=" * 60)
        body = {"prompt": prompt, "max_tokens": max_tokens}
        if temperature is not None:
            body["temperature"] = temperature
        try:
            r = self._rq.post(f"{self.api_url}/generate", json=body, headers=self.headers, timeout=self.timeout)
            if r.status_code != 200:
                self._say(f"\nHTTP Error {r.status_code}")
                return None
            d = r.json()
            if "error" in d:                                  # Test.py:77
                self._say(f"\nError: {d['error']}")
                return None
            self._say(f"\nResponse: {d.get('response', 'No response')}")
            if "time_taken" in d:
                self._say("\nPerformance:")
                self._say(f"   Time: {d.get('time_taken', 'N/A')}")
                self._say(f"   Tokens: {d.get('tokens_generated', 'N/A')}")
                self._say(f"   Speed: {d.get('tokens_per_sec', 'N/A')} tok/s")
            self._say("=" * 60 + "\n")
            return d
        except self._rq.exceptions.Timeout:
            self._say(f"\nRequest timed out after {self.timeout:.0f} seconds!")
            return None
        except Exception as e:
            self._say(f"\nError: {e}")
            return None

    def stream(self, prompt: str, max_tokens: int = 20, temperature: Optional[float] = None):
        """Generator over the server-sent events of `POST /generate` with `"stream": true`: yields each
        new piece of text as it is decoded; the final result dictionary (the one `generate` returns) is
        left in `self.last_result`. Closing the generator early closes the connection, which cancels
        the request on the server."""
        import json
        body = {"prompt": prompt, "max_tokens": max_tokens, "stream": True}
        if temperature is not None:
            body["temperature"] = temperature
        self.last_result = None
        with self._rq.post(f"{self.api_url}/generate", json=body, headers=self.headers, timeout=self.timeout,
                           stream=True) as r:
            if r.status_code != 200:
                raise RuntimeError(f"HTTP Error {r.status_code}")
            for line in r.iter_lines():
                if not line or not line.startswith(b"data: "):
                    continue
                payload = line[6:]
                if payload == b"[DONE]":
                    break
                d = json.loads(payload)
                if "text" in d and "status" not in d:
                    yield d["text"]
                else:
                    self.last_result = d

    def generate_streaming(self, prompt: str, max_tokens: int = 20, temperature: Optional[float] = None) -> Optional[Dict]:
        """`generate`, printing the text while it arrives."""
        self._say("\n" + "=" * 60 + f"\nPrompt: {prompt}\n" + "=" * 60 + "\n\nResponse: ", end="")
        try:
            for piece in self.stream(prompt, max_tokens, temperature):
                if self.verbose:
                    print(piece, end="", flush=True)
        except Exception as e:
            self._say(f"\nError: {e}")
            return None
        d = self.last_result
        if not d or "error" in d:
            self._say(f"\nError: {(d or {}).get('error', 'no result')}")
            return None
        self._say(f"\n\nPerformance:\n   Time: {d.get('time_taken', 'N/A')}\n   Tokens: {d.get('tokens_generated', 'N/A')}"
                  f"\n   Speed: {d.get('tokens_per_sec', 'N/A')} tok/s\n" + "=" * 60 + "\n")
        return d

    def interactive_chat(self, input_fn=input, max_tokens: int = 15):
        self._say("\n" + "=" * 60 + "\nINTERACTIVE CHAT MODE\n" + "=" * 60)
        self._say("Commands:\n  - Type your question and press Enter\n  - 'quit' / 'exit' / 'q' - Exit\n"
                  "  - 'workers' - Check worker status\n  - 'health' - Check API health\n" + "=" * 60 + "\n")
        while True:
            try:
                user_input = input_fn("You: ").strip()
                if not user_input:
                    continue
                low = user_input.lower()
                if low in ("quit", "exit", "q"):
                    self._say("\nGoodbye!")
                    break
                if low == "workers":
                    self.check_workers()
                    continue
                if low == "health":
                    self.check_health()
                    continue
                if getattr(self, "streaming", False):
                    self.generate_streaming(user_input, max_tokens=max_tokens)
                else:
                    self.generate(user_input, max_tokens=max_tokens)    # Test.py:138
            except (KeyboardInterrupt, EOFError):
                self._say("\n\nGoodbye!")
                break
            except Exception as e:
                self._say(f"\nError: {e}\n")


def main(argv=None, input_fn=input):
    """The reference's start menu (Test.py:147-187)."""
    argv = sys.argv[1:] if argv is None else argv
    flags = [x for x in argv if x.startswith("--")]
    argv = [x for x in argv if not x.startswith("--")]
    api_url = argv[0] if argv else input_fn("Enter the API URL: ").strip()
    if not api_url:
        print("No URL provided!")
        return 1
    client = DistributedLLMClient(api_url)
    client.streaming = "--stream" in flags             # print the text while it is decoded
    client.check_workers()
    print("\n" + "=" * 60 + "\nWhat would you like to do?\n1. Interactive chat (recommended)\n2. Single query\n3. Quick test\n" + "=" * 60)
    choice = input_fn("\nChoice (1/2/3): ").strip()
    if choice == "2":
        q = input_fn("\nYour question: ").strip()
        if q:
            client.generate(q, max_tokens=20)
    elif choice == "3":
        print("\nQuick test with short prompt...\n")
        client.generate("Hello", max_tokens=15)
    else:
        client.interactive_chat(input_fn)
    return 0


if __name__ == "__main__":
    sys.exit(main())
