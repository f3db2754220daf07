"""not gpu: every kprn_set_option key the engine accepts is described in include/kprn.h, and every key the header's option list names is accepted
(the header is what a maintainer of the reference-side binding reads: INTEGRATION.md)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_option_keys_in_code_and_header_agree():
    api = open(os.path.join(ROOT, "kprn_amd", "csrc", "kprn_api.hip")).read()
    hdr = open(os.path.join(ROOT, "include", "kprn.h")).read()
    start = api.index("int kprn_set_option(")
    body = api[start:api.index("API_END(h)", start)]
    accepted = set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', body))
    assert len(accepted) >= 20, accepted
    # the header's description of kprn_set_option: the comment block that ends at its declaration
    decl = hdr.index("int kprn_set_option(")
    block = hdr[hdr.rindex("/*", 0, decl):decl]
    # a quoted word is a key where it opens a list entry (" *   "key" ...), continues one ("a" / "b" / "c", "a", "b" ...) or sits in the "(also: ...)" line
    named = set()
    for line in block.splitlines():
        m = re.match(r'\s*\*\s{1,4}"([a-z0-9_]+)"(.*)', line)
        if m:
            named.add(m.group(1))
            rest = m.group(2)
            while True:   # `"a" "0" | "1", "b" "0" | "1"` and `"a" / "b"`: further keys of the same entry follow a comma or a slash
                m2 = re.match(r'[^,/]*[,/]\s*"([a-z0-9_]+)"(.*)', rest)
                if not m2:
                    break
                if m2.group(1) in accepted:
                    named.add(m2.group(1))
                rest = m2.group(2)
        if "(also:" in line or named and line.strip().startswith('*    "'):
            named.update(k for k in re.findall(r'"([a-z0-9_]+)"', line) if k in accepted)
    missing_in_header = sorted(k for k in accepted if '"%s"' % k not in block)
    assert not missing_in_header, "accepted by kprn_set_option but not described in include/kprn.h: %s" % missing_in_header
    unknown = sorted(k for k in named if k not in accepted)
    assert not unknown, "described in include/kprn.h as an option but not accepted by kprn_set_option: %s" % unknown
