#!/usr/bin/env python3
"""Writes tests/golden/glob_kat.json: the glob vectors the reference's own unit tests hold
(crates/matchy-paraglob/src/glob.rs:464-705, `GlobPattern::matches`), transcribed as data — pattern, match mode, texts that must match,
texts that must not — for the cases where a DATABASE holding the pattern as its only key answers a query exactly like
`GlobPattern::matches` answers: keys without wildcard are literal entries (exact match), globs reach the verifier when they carry a
literal of >= 3 bytes (the Aho-Corasick prefilter: paraglob_offset.rs:553-555) or none at all (pure wildcards are always verified:
:1089-1134). Left out on purpose, with the reason: escaped wildcards (`file\\*.txt`: the database layer classifies the key as a
paraglob LITERAL, backslash included — paraglob_offset.rs:93-108 — so the database does not behave like GlobPattern there), globs
whose every literal is shorter than 3 bytes (`*Ż*`, `*a*b*c*…`: they can never match through a database), the empty
pattern (not a valid key) and the four invalid patterns (mmdb_builder.rs:420-428 turns an invalid glob into a literal key)."""
import json
from pathlib import Path

C = []


def case(ref, pattern, match, nomatch=(), ci=False):
    C.append({"ref": ref, "pattern": pattern, "case_insensitive": ci, "match": list(match), "nomatch": list(nomatch)})


case("glob.rs:465", "hello", ["hello"], ["hello world", "Hell o", ""])
case("glob.rs:474", "hello", ["hello", "HELLO", "HeLLo"], ["hello world"], ci=True)
case("glob.rs:483", "*.txt", [".txt", "file.txt", "my.file.txt"], ["file.pdf", "txt"])
case("glob.rs:493", "hello*world", ["helloworld", "hello world", "hello beautiful world"], ["hello", "world", "goodbye world"])
case("glob.rs:504", "*hello*world*", ["hello world", "say hello to the world today", "helloworld"], ["hello", "world"])
case("glob.rs:514", "file?.txt", ["file1.txt", "fileA.txt", "file?.txt"], ["file.txt", "file10.txt"])
case("glob.rs:524", "???", ["abc", "123"], ["ab", "abcd"])
case("glob.rs:533", "file[123].txt", ["file1.txt", "file2.txt", "file3.txt"], ["file4.txt", "fileA.txt"])
case("glob.rs:543", "file[0-9].txt", ["file0.txt", "file5.txt", "file9.txt"], ["fileA.txt"])
case("glob.rs:552", "[a-zA-Z]", ["a", "z", "A", "Z"], ["0", "!"])
case("glob.rs:563", "file[!0-9].txt", ["fileA.txt", "file_.txt"], ["file0.txt", "file9.txt"])
case("glob.rs:572", "[^abc]", ["d", "z"], ["a", "b"])
case("glob.rs:596", "**/[a-z]*.{txt,md}", ["some/path/file.{txt,md}"])
case("glob.rs:610", "*", ["", "anything", "multiple words"])
case("glob.rs:642", "[a-z]", ["a", "A", "z", "Z"], ci=True)
case("glob.rs:651", "hello*", ["hello世界", "hello🌍"])
case("glob.rs:676", "*世*界*", ["hello世foo界bar"])

out = Path(__file__).with_name("glob_kat.json")
out.write_text(json.dumps({"cases": C}, indent=1, ensure_ascii=False) + "\n")
print(len(C), "cases ->", out)
