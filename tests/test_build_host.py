"""CPU: the source hash that decides rebuilds (geomconsistentfr_amd/build.py) -- what the comment stripper must not mis-read."""
from geomconsistentfr_amd import build as hb


def test_comment_edits_do_not_change_the_code_and_code_edits_do():
    a = "int f(int x) { return x + 1; }  // one\n/* block */\nint g();\n"
    b = "int f(int x) { return x + 1; }  // another comment\n\n/* other\n block */\nint g();   \n"
    c = "int f(int x) { return x + 2; }  // one\n/* block */\nint g();\n"
    assert hb._code_only(a) == hb._code_only(b)
    assert hb._code_only(a) != hb._code_only(c)


def test_digit_separators_are_not_character_literals():
    """advisor r03: 1'000 used to open a 'character literal' that swallowed code (and later comments) up to the next apostrophe"""
    a = "constexpr int N = 1'000'000; // count\nint a = 1; /* x */ int b = 'c';\n"
    b = "constexpr int N = 1'000'000; // count\nint a = 2; /* x */ int b = 'c';\n"
    ca, cb = hb._code_only(a), hb._code_only(b)
    assert ca != cb
    assert "count" not in ca and "/*" not in ca and "1'000'000" in ca and "'c'" in ca
    assert hb._code_only("x = 0x7f'ff; // c\ny = 1;") == "x = 0x7f'ff;\ny = 1;"


def test_raw_strings_are_copied_verbatim():
    a = 'const char *s = R"gc(// not a comment " /* nor this */ )gc"; // real comment\nint z;\n'
    ca = hb._code_only(a)
    assert "// not a comment" in ca and "/* nor this */" in ca and "real comment" not in ca and "int z;" in ca
    assert hb._code_only('auto FOOR"x"; // c') == 'auto FOOR"x";'      # an identifier ending in R is not a raw-string prefix


def test_recorded_hash_file_has_code_hash_first():
    code, full = hb.recorded_hashes()
    assert code is None or len(code) == 64
    assert full is None or len(full) == 64
