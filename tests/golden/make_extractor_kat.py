#!/usr/bin/env python3
"""Writes tests/golden/extractor_kat.json.

Known-answer vectors for the tokenizer, transcribed (inputs and asserted outcomes only, no code) from the
assertions of the reference's own unit tests in crates/matchy-extractor/src/lib.rs:1976-3626 and
crates/matchy/src/processing/mod.rs:615-702. `ref` gives the reference line of the test that pins the case.
`expect` lists, per item type, the extracted texts in input order (IPs in canonical Display form); a type that
is absent from `expect` is not asserted by the reference test and is not checked. `total`, when present, is the
asserted total number of matches.
"""
import json
from pathlib import Path

MD5 = "5d41402abc4b2a76b9719d911017c592"
SHA1 = "2fd4e1c67a2d28fced849ee1bb76e7391b93eb12"
SHA256 = "2c26b46b68ffc68ff99b453c1d30413413422d706483bfa0f98a5e886266e7ae"
SHA384 = "cb00753f45a35e8bb5a03d699ac65007272c32ab0eded1631a8b605a43ff5bed8086072ba1e7cc2358baeca134c825a7"
SHA512 = ("cf83e1357eefb8bdf1542850d66d8007d620e4050b5715dc83f4a921d36ce9ce"
          "47d0d13c5d85f2b0ff8318d2877eec2f63b931bd47417a81a538327af927da3e")
BTC1 = "1A1zP1eP5QGefi2DMPTfTL5SLmv7DivfNa"
BTC3 = "3Cbq7aT1tY8kMxWLbitaG7yT6bPbKChq64"
BECH = "bc1qar0srrr7xfkvy5l643lydnw9re59gtzzwf5mdq"
ETH_LO = "0x5aeda56215b167893e80b4fe645ba6d5bab767de"
ETH_CK = "0x5aAeb6053F3E94C9b9A09f33669435E7Ef1BeAed"

C = []


def case(ref, text, expect, total=None, flags=255, min_labels=2, raw=None):
    d = {"ref": ref, "flags": flags, "min_labels": min_labels, "expect": expect}
    if raw is not None:
        d["input_hex"] = raw.hex()
    else:
        d["input"] = text
    if total is not None:
        d["total"] = total
    C.append(d)


# --- domains
case("lib.rs:2093", "Visit example.com for more info", {"Domain": ["example.com"]}, total=1)
case("lib.rs:2108", "Check google.com and github.com", {"Domain": ["google.com", "github.com"]}, total=2)
case("lib.rs:2119", "Visit api.example.com today", {"Domain": ["api.example.com"]}, total=1)
case("lib.rs:2130", "Go to https://www.example.com/path", {"Domain": ["www.example.com"]}, total=1)
case("lib.rs:2143", "Visit example.com and api.test.example.com", {"Domain": ["api.test.example.com"]}, total=1,
     flags=1, min_labels=3)
case("lib.rs:2345", "Request: host=api.example.com method=GET path=/test", {"Domain": ["api.example.com"]})
case("lib.rs:2877", "Visit .app or .com for info", {"Domain": []})
case("lib.rs:2854", "Invalid domain: Kagi%20Assistant.app", {"Domain": ["20Assistant.app"]} if False else {})
case("lib.rs:2222", "Visit münchen.de for info", {"Domain": ["münchen.de"]}, total=1)
case("lib.rs:2297", None, {"Domain": []}, raw=b"Visit \xff\xc0.com")
case("lib.rs:2262", None, {"Domain": ["evil.com"]}, raw=b"Log: \xff\xfe evil.com \x80")
case("lib.rs:2322", "This is blah.community stuff", {})
# --- IPv4
case("lib.rs:2182", "Server at 192.168.1.1 responded", {"IPv4": ["192.168.1.1"]})
case("lib.rs:2200", "Traffic from 10.0.0.5 to 172.16.0.10", {"IPv4": ["10.0.0.5", "172.16.0.10"]})
case("lib.rs:2368", "Not IPs: 256.1.1.1 1.2.3.999 1.2.3", {"IPv4": []})
case("lib.rs:2629", "Invalid IP: 2025.36.0.72591908", {"IPv4": []})
case("lib.rs:2648", "Invalid IP: 460.1.1.2", {"IPv4": []})
case("lib.rs:2667", "Invalid IP: 26.0..26.0", {"IPv4": []})
case("lib.rs:2387", "Request from 10.1.2.3 to api.example.com at 192.168.1.100",
     {"IPv4": ["10.1.2.3", "192.168.1.100"], "Domain": ["api.example.com"]})
# --- e-mail
case("lib.rs:2416", "Contact user@example.com for info", {"Email": ["user@example.com"]})
case("lib.rs:2435", "Email alice@test.com or bob@company.org", {"Email": ["alice@test.com", "bob@company.org"]})
case("lib.rs:2455", "Send to user+tag@example.com", {"Email": ["user+tag@example.com"]})
case("lib.rs:2474", "2024-01-15 user@example.com from 10.1.2.3 accessed api.test.com",
     {"Email": ["user@example.com"], "IPv4": ["10.1.2.3"], "Domain": ["example.com", "api.test.com"]})
case("lib.rs:2686", "Invalid email: s...@example.com", {"Email": []})
case("lib.rs:2709", "Invalid email: .@example.com", {"Email": []})
case("lib.rs:2732", "Valid email: 34480FE2-5610-4973-AA09-3ABB60D38D55@example.com",
     {"Email": ["34480FE2-5610-4973-AA09-3ABB60D38D55@example.com"]})
case("lib.rs:2759", "Invalid email: user@192.168.1.222", {"Email": []})
case("lib.rs:2782", "Invalid email: test@Uv3.peer", {"Email": []})
# --- IPv6
case("lib.rs:2517", "Server at 2001:db8:85a3::8a2e:370:7334 responded", {"IPv6": ["2001:db8:85a3::8a2e:370:7334"]})
case("lib.rs:2537", "Connecting to 2001:db8::1", {"IPv6": ["2001:db8::1"]})
case("lib.rs:2556", "Address 2001:0db8::1 connects to 2606:2800:220:1::248", {"IPv6": ["2001:db8::1", "2606:2800:220:1::248"]})
case("lib.rs:2578", "Traffic from 2001:db8::1 to 2001:db8::2", {"IPv6": ["2001:db8::1", "2001:db8::2"]})
case("lib.rs:2598", "IPv4: 192.168.1.1 IPv6: 2001:db8::1", {"IPv4": ["192.168.1.1"], "IPv6": ["2001:db8::1"]})
for t in ("Tiny IPv6: e::f", "Tiny IPv6: ce::A", "Tiny IPv6: e::add"):
    case("lib.rs:2801", t, {"IPv6": []})
case("lib.rs:2831", "Invalid IPv6: FEC0050519FB::c", {"IPv6": []})
case("lib.rs:2854", "Invalid IPv6: 7::31BD71E4", {"IPv6": []})
case("lib.rs:2924", "Link-local address: fe80::1 and fe80::dead:beef", {"IPv6": []})
# --- hashes
case("lib.rs:2949", f"File hash: {MD5} uploaded", {"MD5": [MD5]})
case("lib.rs:2969", f"SHA1: {SHA1} verified", {"SHA1": [SHA1]})
case("lib.rs:2989", f"SHA256: {SHA256} detected", {"SHA256": [SHA256]})
case("lib.rs:3013", f"SHA384: {SHA384} verified", {"SHA384": [SHA384]})
case("lib.rs:2038", f"SHA512: {SHA512} found", {"SHA512": [SHA512]})
case("lib.rs:3033", f"MD5: {MD5} SHA1: {SHA1}", {"MD5": [MD5], "SHA1": [SHA1]})
case("lib.rs:3054", "Hash: 5D41402ABC4B2A76B9719D911017C592 found", {"MD5": ["5D41402ABC4B2A76B9719D911017C592"]})
case("lib.rs:3073", "Hash: 5d41402AbC4b2A76b9719D911017c592 mixed", {"MD5": ["5d41402AbC4b2A76b9719D911017c592"]})
case("lib.rs:3092", "Hash: 5d41402abc4b2a76b9719d91101 invalid", {"MD5": [], "SHA1": [], "SHA256": [], "SHA384": [], "SHA512": []})
case("lib.rs:3111", "Hash: 5d41402abc4b2a76b9719d911017c5gz invalid", {"MD5": [], "SHA1": [], "SHA256": [], "SHA384": [], "SHA512": []})
case("lib.rs:3130", f"Hash: [{MD5}] in brackets", {"MD5": [MD5]})
case("lib.rs:3149", f"2024-01-15 malware.exe MD5={MD5} detected from 192.168.1.100", {"MD5": [MD5], "IPv4": ["192.168.1.100"]})
case("lib.rs:3178", f"Line1: {MD5}\nLine2: {SHA1}\n", {"MD5": [MD5], "SHA1": [SHA1]})
case("lib.rs:3198", f"Hash: {MD5} should not extract", {"MD5": []}, flags=255 & ~16)
case("lib.rs:3218", "UUID: 550e8400-e29b-41d4-a716-446655440000 not a hash", {"MD5": [], "SHA1": [], "SHA256": [], "SHA384": [], "SHA512": []})
# --- crypto
case("lib.rs:3240", f"Send to {BTC1} for payment", {"Bitcoin": [BTC1]})
case("lib.rs:3260", f"Payment to {BTC3} confirmed", {"Bitcoin": [BTC3]})
case("lib.rs:3280", f"Withdraw to {BECH}", {"Bitcoin": [BECH]})
case("lib.rs:3300", "Fake address 1A1zP1eP5QGefi2DMPTfTL5SLmv7Divf00 is invalid", {"Bitcoin": []})
case("lib.rs:3324", "Short address 1A1zP1eP is invalid", {"Bitcoin": []})
case("lib.rs:3343", f"Transfer to {ETH_LO}", {"Ethereum": [ETH_LO]})
case("lib.rs:3363", f"Send to {ETH_CK}", {"Ethereum": [ETH_CK]})
case("lib.rs:3383", "Bad address 0x5aAeb6053f3e94c9b9a09f33669435e7ef1beaed", {"Ethereum": []})
case("lib.rs:3407", "Short address 0x5aeda56215b167893e80b4fe645ba6d5bab7", {"Ethereum": []})
case("lib.rs:3426", "Invalid 0x5aeda56215b167893e80b4fe645ba6d5bab767dg", {"Ethereum": []})
case("lib.rs:3478", "Fake 1AdUndXHHZ6cfufTMvppY6JwXNouMBzSkbLYfpAV5Usx3skxNgYeYTRj5UzqtReoS44qo9mtmXCqY45DJ852K5Jv2684Rge", {"Monero": []})
case("lib.rs:3501", "Short 4AdUndXHHZ6cfufTMvppY6JwXNouMBzSkbLYfpAV5Usx", {"Monero": []})
case("lib.rs:3520", f"Transaction from 192.168.1.1 to {BECH} via example.com",
     {"IPv4": ["192.168.1.1"], "Domain": ["example.com"], "Bitcoin": [BECH]})
case("lib.rs:3559", f"Send to {BTC1} or {ETH_LO}", {"Bitcoin": [], "Ethereum": [], "Monero": []}, flags=255 & ~(32 | 64 | 128))
case("lib.rs:3588", f"2025-01-15 10:32:45 Transaction to={ETH_LO} value=1000000000000000000", {"Ethereum": [ETH_LO]})
case("lib.rs:3608", f"Line1: {BTC1}\nLine2: {BTC3}\n", {"Bitcoin": [BTC1, BTC3]})
# --- Worker tests (processing/mod.rs:615-702): extraction side
case("processing/mod.rs:630", "Connection from 1.2.3.4 detected", {"IPv4": ["1.2.3.4"]})
case("processing/mod.rs:672", "DNS query to evil.com from 8.8.8.8", {"IPv4": ["8.8.8.8"], "Domain": ["evil.com"]})
# --- chunk doc-test (lib.rs:405)
case("lib.rs:405", "test@example.com\n192.168.1.1\nmalware.com", {"Email": ["test@example.com"], "IPv4": ["192.168.1.1"]})

# drop the placeholder with an empty expectation that asserts nothing about a specific text
C = [c for c in C if not (c.get("input") == "Invalid domain: Kagi%20Assistant.app")]
# lib.rs:2854-2875 asserts only that no extracted domain contains '%'
C.append({"ref": "lib.rs:2854", "flags": 255, "min_labels": 2, "input": "Invalid domain: Kagi%20Assistant.app",
          "expect": {}, "forbid_substring": {"Domain": "%"}})
C.append({"ref": "lib.rs:2322", "flags": 255, "min_labels": 2, "input": "This is blah.community stuff",
          "expect": {}, "forbid_suffix": {"Domain": ".com"}})

out = Path(__file__).with_name("extractor_kat.json")
out.write_text(json.dumps({"cases": C}, indent=1, ensure_ascii=True) + "\n")
print(len(C), "cases ->", out)
