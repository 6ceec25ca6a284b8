// q3_cli -- `qwen3 inference` (qwen3-cli/src/main.rs:36-93) on the MI355X engine, host side in C++ over the C ABI of
// include/qwen3_hip.h.  The same flags, modes and loops as the reference:
//
//     q3_cli inference <checkpoint> [-t 1.0] [-p 0.9] [-s SEED] [-c CTX] [-m generate|chat] [-i INPUT] [-y SYSTEM] [-r 0|1]
//
//   * tokenizer: `<checkpoint>.tokenizer` + `.template*` files (qwen3-inference/src/tokenizer.rs:28-237): byte-level BPE, greedy
//     leftmost best-score merges, `<...>` special tokens; the reference scans the vocabulary for every lookup (O(V)), here a hash map
//     from bytes to the FIRST index with those bytes gives the same answers;
//   * `generate` (generation.rs:9-48): the prompt is echoed, decoding starts from its last token over a zero KV prefix;
//   * `chat` (generation.rs:50-151): template rendering, every prompt token forwarded (one rng coin each), decode until BOS / EOS,
//     the window wraps to position 0 without clearing the cache (generation.rs:65-69);
//   * forward, prompt loop and sampling (temperature / top-p / xorshift64*) run on the device: q3_prefill_batched / q3_prefill,
//     q3_forward_argmax (which draws with the device sampler once q3_sampler_set was called with temperature > 0).
// `export` stays with the Python tool (python -m qwen3_rs_amd.cli export): it is offline and needs a JSON / safetensors reader.
// Build: make -C qwen3-rs_amd cli   (g++, links libqwen3_hip.so).  The Python CLI (qwen3_rs_amd/cli.py) is the same program; the
// tests run both on the same checkpoint and compare the bytes they print.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <chrono>
#include <iostream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/qwen3_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// tokenizer.rs:28-101: u32 max_token_length, u32 bos, u32 eos, then per token f32 score, u32 len, bytes; records missing at the
// end of the file become empty tokens with score 0
// ---------------------------------------------------------------------------------------------------------------
struct Tokenizer {
    std::vector<std::string> vocab;
    std::vector<float> scores;
    std::unordered_map<std::string, int> first;      // bytes -> first index (Iterator::position, tokenizer.rs:145-151)
    uint32_t max_token_length = 0, bos = 0, eos = 0;
    std::string prompt_template, system_prompt_template;

    static bool read_file(const std::string& path, std::string& out) {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) return false;
        char buf[1 << 16];
        size_t n;
        out.clear();
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
        fclose(f);
        return true;
    }
    // tokenizer.rs:103-122
    static std::string load_template(const std::string& ckpt, bool with_system, bool thinking) {
        const char* suffix = with_system ? (thinking ? ".template.with-system-and-thinking" : ".template.with-system")
                                         : (thinking ? ".template.with-thinking" : ".template");
        std::string s;
        if (!read_file(ckpt + suffix, s)) {
            fprintf(stderr, "Warning: Could not load prompt template %s%s\n", ckpt.c_str(), suffix);
            return "";
        }
        return s;
    }
    bool load(const std::string& ckpt, int vocab_size, bool thinking) {
        std::string d;
        if (!read_file(ckpt + ".tokenizer", d)) {
            fprintf(stderr, "Error: cannot open %s.tokenizer\n", ckpt.c_str());
            return false;
        }
        if (d.size() < 12) {
            fprintf(stderr, "Error: %s.tokenizer is truncated\n", ckpt.c_str());
            return false;
        }
        memcpy(&max_token_length, d.data(), 4);
        memcpy(&bos, d.data() + 4, 4);
        memcpy(&eos, d.data() + 8, 4);
        size_t off = 12;
        vocab.assign((size_t)vocab_size, std::string());
        scores.assign((size_t)vocab_size, 0.0f);
        for (int i = 0; i < vocab_size; ++i) {
            if (off + 4 > d.size()) continue;
            memcpy(&scores[(size_t)i], d.data() + off, 4);
            off += 4;
            if (off + 4 > d.size()) continue;
            uint32_t n;
            memcpy(&n, d.data() + off, 4);
            off += 4;
            if (off + n > d.size()) { off = d.size(); continue; }
            vocab[(size_t)i].assign(d.data() + off, n);
            off += n;
        }
        for (int i = 0; i < vocab_size; ++i) first.emplace(vocab[(size_t)i], i);      // (emplace keeps the first)
        prompt_template = load_template(ckpt, false, thinking);
        system_prompt_template = load_template(ckpt, true, thinking);
        return true;
    }
    const std::string& decode(int token) const {
        static const std::string empty;
        return token >= 0 && (size_t)token < vocab.size() ? vocab[(size_t)token] : empty;
    }
    int lookup(const std::string& bytes) const {
        auto it = first.find(bytes);
        return it == first.end() ? -1 : it->second;
    }
    // the text as characters (tokenizer.rs:165 `text.chars()`): byte ranges of the UTF-8 code points; a malformed byte stands alone
    static std::vector<std::string> chars_of(const std::string& s) {
        std::vector<std::string> out;
        for (size_t i = 0; i < s.size();) {
            const unsigned char c = (unsigned char)s[i];
            size_t n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1;
            if (i + n > s.size()) n = 1;
            for (size_t k = 1; k < n; ++k)
                if (((unsigned char)s[i + k] & 0xc0) != 0x80) { n = 1; break; }
            out.emplace_back(s, i, n);
            i += n;
        }
        return out;
    }
    // tokenizer.rs:165-237
    std::vector<int> encode(const std::string& text) const {
        std::vector<int> tokens;
        const std::vector<std::string> ch = chars_of(text);
        for (size_t i = 0; i < ch.size();) {
            bool found = false;
            if (ch[i] == "<") {
                const size_t limit = std::min(ch.size(), i + (size_t)max_token_length);
                size_t end = 0;
                for (size_t j = i + 1; j < limit; ++j)
                    if (ch[j] == ">") { end = j; break; }
                if (end) {
                    std::string special;
                    for (size_t j = i; j <= end; ++j) special += ch[j];
                    const int id = lookup(special);
                    if (id >= 0) {
                        tokens.push_back(id);
                        i = end + 1;
                        found = true;
                    }
                }
            }
            if (!found) {
                const int id = lookup(ch[i]);
                if (id >= 0) tokens.push_back(id);
                else printf("Warning: unknown character '%s' in input, skipping.\n", ch[i].c_str());
                ++i;
            }
        }
        for (;;) {      // merge the leftmost best-scoring pair until none is left (tokenizer.rs:208-234)
            float best_score = -1e10f;
            int best_id = -1;
            size_t best_idx = 0;
            for (size_t k = 0; k + 1 < tokens.size(); ++k) {
                const int id = lookup(vocab[(size_t)tokens[k]] + vocab[(size_t)tokens[k + 1]]);
                if (id >= 0 && scores[(size_t)id] > best_score) {
                    best_score = scores[(size_t)id];
                    best_id = id;
                    best_idx = k;
                }
            }
            if (best_id < 0) break;
            tokens[best_idx] = best_id;
            tokens.erase(tokens.begin() + (long)best_idx + 1);
        }
        return tokens;
    }
    static std::string replace_all(const std::string& tmpl, const std::string& with) {
        std::string out;
        size_t p = 0, q;
        while ((q = tmpl.find("%s", p)) != std::string::npos) {
            out.append(tmpl, p, q - p);
            out += with;
            p = q + 2;
        }
        out.append(tmpl, p, std::string::npos);
        return out;
    }
    // generation.rs:188-195
    std::string render_prompt(size_t pos, const std::string* system_prompt, const std::string& user) const {
        if (pos == 0 && system_prompt != nullptr) return replace_all(system_prompt_template, *system_prompt + "\n" + user);
        return replace_all(prompt_template, user);
    }
};

void out_bytes(const std::string& s) {
    fwrite(s.data(), 1, s.size(), stdout);
    fflush(stdout);
}

struct Metrics {      // TokenMetrics, generation.rs:198-233
    long n = 0;
    bool started = false;
    std::chrono::steady_clock::time_point t0;
    void start() { if (!started) { started = true; t0 = std::chrono::steady_clock::now(); } }
    void report_and_reset() {
        if (started && n) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "\n[%.2f tk/s, %ld tokens in %.2fs]\n", n / dt, n, dt);
        }
        n = 0;
        started = false;
    }
};

int fail_engine(const char* what) {
    fprintf(stderr, "Error: %s: %s\n", what, q3_last_error());
    return 1;
}

// generation.rs:9-48
int run_generate(q3_engine* e, const Tokenizer& tok, int seq_len, const std::string* prompt) {
    const std::vector<int> pt = tok.encode(prompt ? *prompt : std::string());
    if (pt.empty()) {
        fprintf(stderr, "Please provide a prompt\n");
        return 1;
    }
    for (size_t i = 0; i + 1 < pt.size() && i < (size_t)seq_len; ++i) out_bytes(tok.decode(pt[i]));     // echoed, never forwarded
    int token = pt.back();
    size_t pos = pt.size() - 1;
    Metrics m;
    while (pos < (size_t)seq_len) {
        m.start();
        int32_t nxt = -1;
        if (q3_forward_argmax(e, (size_t)token, pos, &nxt) != Q3_OK) return fail_engine("forward");
        ++m.n;
        if ((uint32_t)nxt == tok.bos || (uint32_t)nxt == tok.eos) break;
        out_bytes(tok.decode(token));
        token = nxt;
        ++pos;
    }
    m.report_and_reset();
    out_bytes("\n");
    return 0;
}

// 2,048 positions per weight pass where the checkpoint's shape allows it, else the sequential device loop: same result
int prefill(q3_engine* e, const std::vector<int>& ids, size_t pos, int32_t* next) {
    std::vector<int32_t> t(ids.begin(), ids.end());
    const char* off = getenv("Q3_CLI_BATCHED_PREFILL");
    if (!(off && strcmp(off, "0") == 0)) {
        const int rc = q3_prefill_batched(e, t.data(), t.size(), pos, next);
        if (rc == Q3_OK) return rc;
        if (rc != Q3_ERR_UNSUPPORTED) return rc;
    }
    return q3_prefill(e, t.data(), t.size(), pos, next);
}

// generation.rs:50-151, loop for loop (see qwen3_rs_amd/cli.py::run_chat)
int run_chat(q3_engine* e, const Tokenizer& tok, int seq_len, const std::string* cli_prompt, const std::string* system_prompt) {
    size_t pos = 0;
    bool user_turn = true;
    int32_t nxt = 0;
    Metrics m;
    for (;;) {
        if (pos >= (size_t)seq_len) {      // "Reset context if window exceeded": the cache is not cleared
            pos = 0;
            user_turn = true;
            out_bytes("\n");
        }
        if (user_turn) {
            m.report_and_reset();
            std::string user;
            if (pos == 0 && cli_prompt) user = *cli_prompt;
            else if (cli_prompt) user.clear();
            else {
                out_bytes("> ");
                if (!std::getline(std::cin, user)) user.clear();
                const size_t a = user.find_first_not_of(" \t\r\n\v\f"), b = user.find_last_not_of(" \t\r\n\v\f");
                user = a == std::string::npos ? std::string() : user.substr(a, b - a + 1);
            }
            if (user.empty() && !(pos == 0 && cli_prompt)) break;
            std::vector<int> ids = tok.encode(tok.render_prompt(pos, system_prompt, user));
            const size_t room = (size_t)seq_len > pos ? (size_t)seq_len - pos : 0;
            if (ids.size() > room) ids.resize(room);
            if (!ids.empty()) {
                if (prefill(e, ids, pos, &nxt) != Q3_OK) return fail_engine("prefill");
                pos += ids.size();
            }
            user_turn = false;
        } else {
            if ((uint32_t)nxt == tok.bos || (uint32_t)nxt == tok.eos) {
                m.report_and_reset();
                out_bytes("\n");
                user_turn = true;
                continue;
            }
            m.start();
            out_bytes(tok.decode(nxt));
            int32_t n2 = -1;
            if (q3_forward_argmax(e, (size_t)nxt, pos, &n2) != Q3_OK) return fail_engine("forward");
            nxt = n2;
            ++m.n;
            ++pos;
        }
    }
    return 0;
}

void usage() {
    fprintf(stderr,
            "Qwen3 inference on the MI355X engine\n\n"
            "Usage: q3_cli inference <checkpoint> [-t TEMPERATURE] [-p TOPP] [-s SEED] [-c CONTEXT] [-m generate|chat]\n"
            "                                     [-i INPUT] [-y SYSTEM] [-r 0|1]\n"
            "  -t, --temperature  [0, inf)   default 1.0\n  -p, --topp         [0, 1]     default 0.9\n"
            "  -s, --seed         random seed (default: time)\n  -c, --context      context window size (default: the checkpoint's)\n"
            "  -m, --mode         generate | chat (default chat)\n  -i, --input        input prompt\n"
            "  -y, --system       system prompt (chat mode)\n  -r, --reasoning    0 = no thinking, 1 = thinking (default 0)\n");
}

}  // namespace

int main(int argc, char** argv) {
    // (test hook, no engine and no GPU: q3_cli tokenize <checkpoint> <vocab_size> <text> prints the ids of `text`, then the bytes they decode to)
    if (argc == 5 && strcmp(argv[1], "tokenize") == 0) {
        Tokenizer tok;
        if (!tok.load(argv[2], atoi(argv[3]), false)) return 1;
        const std::vector<int> ids = tok.encode(argv[4]);
        std::string line, bytes;
        for (size_t i = 0; i < ids.size(); ++i) { line += (i ? " " : "") + std::to_string(ids[i]); bytes += tok.decode(ids[i]); }
        out_bytes(line + "\n" + bytes);
        return 0;
    }
    // (test hook: q3_cli render <checkpoint> <vocab_size> <pos> <system prompt or "-"> <user prompt> prints the rendered turn)
    if (argc == 7 && strcmp(argv[1], "render") == 0) {
        Tokenizer tok;
        if (!tok.load(argv[2], atoi(argv[3]), false)) return 1;
        const std::string sys = argv[5];
        out_bytes(tok.render_prompt((size_t)atol(argv[4]), sys == "-" ? nullptr : &sys, argv[6]));
        return 0;
    }
    if (argc < 3 || strcmp(argv[1], "inference") != 0) {
        usage();
        return 1;
    }
    const std::string ckpt = argv[2];
    double temperature = 1.0, topp = 0.9;
    bool have_seed = false, have_input = false, have_system = false;
    unsigned long long seed = 0;
    long context = 0, reasoning = 0;
    std::string mode = "chat", input, system_prompt;
    for (int i = 3; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) {
                fprintf(stderr, "Error: %s needs a value\n", a.c_str());
                exit(1);
            }
            return argv[++i];
        };
        if (a == "-t" || a == "--temperature") temperature = atof(val());
        else if (a == "-p" || a == "--topp") topp = atof(val());
        else if (a == "-s" || a == "--seed") { seed = strtoull(val(), nullptr, 10); have_seed = true; }
        else if (a == "-c" || a == "--context") context = atol(val());
        else if (a == "-m" || a == "--mode") mode = val();
        else if (a == "-i" || a == "--input") { input = val(); have_input = true; }
        else if (a == "-y" || a == "--system") { system_prompt = val(); have_system = true; }
        else if (a == "-r" || a == "--reasoning") reasoning = atol(val());
        else {
            fprintf(stderr, "Error: unknown argument %s\n", a.c_str());
            usage();
            return 1;
        }
    }
    if (mode != "generate" && mode != "chat") {
        fprintf(stderr, "Error: Unknown mode: %s\n", mode.c_str());
        return 1;
    }
    q3_engine* e = nullptr;
    if (q3_create(ckpt.c_str(), context > 0 ? (uint32_t)context : 0u, 0, 0u, &e) != Q3_OK) return fail_engine("cannot load the checkpoint");
    q3_config cfg;
    if (q3_get_config(e, &cfg) != Q3_OK) return fail_engine("get_config");
    Tokenizer tok;
    if (!tok.load(ckpt, cfg.vocab_size, reasoning != 0)) {
        q3_destroy(e);
        return 1;
    }
    if (!have_seed) seed = (unsigned long long)time(nullptr);      // lib.rs: SystemTime seconds when no seed is given
    const float t = (float)(temperature < 0.0 ? 0.0 : temperature), p = (float)(topp < 0.0 ? 0.0 : (topp > 1.0 ? 1.0 : topp));
    if (q3_sampler_set(e, t, p, (uint64_t)seed) != Q3_OK) return fail_engine("sampler");
    const int rc = mode == "generate" ? run_generate(e, tok, cfg.seq_len, have_input ? &input : nullptr)
                                      : run_chat(e, tok, cfg.seq_len, have_input ? &input : nullptr, have_system ? &system_prompt : nullptr);
    q3_destroy(e);
    return rc;
}
