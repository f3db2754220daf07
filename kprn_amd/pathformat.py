"""Path-set -> `.int` formatting: the step directly in front of the hot path.

Python-3 restatement of release/songPathRnn/data/movie_data_format.py (driver: movie_data_format.sh:2), plus the
three small helpers around it (split_data.py, movie_data_list.py, format_entity_pair.py).  Same inputs
(`<input_dir>/{positive,negative,test}_matrix.tsv.translated`, the five vocab files), same outputs
(`<out_dir>/{train,test}/{train,test}.txt.<numPaths>.int`), same quirks:

  * a line is `e1 \\t e2 \\t path###path###... [\\t label]`; a path is `r-e-r-e-...-r` (relations at even token
    positions, entities at odd ones)                                                    movie_data_format.py:86-96,230-262
  * path length in steps = ntokens // 2 + 2 (Python-2 integer division): one step per relation (paired with the
    entity it leaves from) + the terminal (e2, #END_RELATION) step                       movie_data_format.py:96,245,280-284
  * max_length = min(-m, longest path over ALL three files); longer paths are dropped     movie_data_format.py:83-99,246-249
  * LEFT padding with the all-#PAD_TOKEN feature                                          movie_data_format.py:166-190,250-254
  * step feature = `type ids (numTypes slots), entity id, relation id`, comma separated; unknown names fall back to
    #UNK_ENTITY_TYPE / #UNK_ENTITY / #UNK_RELATION; type ids are sorted AS STRINGS, truncated, reversed, and
    left-padded with #PAD_TOKEN to the slot count                                         movie_data_format.py:102-159
  * train labels: positive file -> "1", negative file -> "-1"; test file carries its label in column 4; mapped through
    vocab/domain-label["domain"]                                                          movie_data_format.py:207-215,234-236,301-303
  * a pair's kept paths are joined with ';', steps with ' '; the pair is appended to the file of its path COUNT
    (bucketing by #paths); pairs with no path left are counted as missed                   movie_data_format.py:295-314
The only/get-only-relation modes (-o / -g) are restated as well.

Parity: pinned against the reference script itself (run under Python 3 with `xrange`/`/` patched in memory) on slices of
its shipped sample inputs: tests/golden/pathformat/ + tests/golden/make_pathformat_golden.py, tests/test_pathformat.py.
"""
import argparse
import json
import os

TRAIN_FILES = ("positive_matrix.tsv.translated", "negative_matrix.tsv.translated", "test_matrix.tsv.translated")


def _read_two_col(path, as_list=False):
    d = {}
    with open(path, "r") as f:
        for line in f:
            parts = line.strip().split("\t")
            d[parts[0]] = [parts[1]] if as_list else parts[1]
    return d


class Vocabs:
    """vocab/{entity_type_id,all_relation_id,all_entity_id,entity_to_type}.txt + vocab/domain-label
    (movie_data_format.py:36-79).  Values stay strings, as in the reference."""

    def __init__(self, vocab_dir, entity_vocab_file="all_entity_id.txt", entity_type_map_file="entity_to_type.txt"):
        self.label2int = json.load(open(os.path.join(vocab_dir, "domain-label"), "r"))
        self.entity = _read_two_col(os.path.join(vocab_dir, entity_vocab_file))
        self.entity_type = _read_two_col(os.path.join(vocab_dir, "entity_type_id.txt"))
        self.entity_type_map = _read_two_col(os.path.join(vocab_dir, entity_type_map_file), as_list=True)
        self.relation = _read_two_col(os.path.join(vocab_dir, "all_relation_id.txt"))


class PathFormatter:
    def __init__(self, vocabs, max_path_length, max_num_types, only_relation=False, get_only_relation=False):
        self.v = vocabs
        self.max_possible = int(max_path_length)
        self.num_type_slots = int(max_num_types)
        self.only_relation = bool(only_relation)
        self.get_only_relation = bool(get_only_relation)
        v = vocabs
        if self.only_relation or self.get_only_relation:  # movie_data_format.py:166-168
            self.pad_feature = str(v.relation["#PAD_TOKEN"])
        else:                                             # :169-176
            self.pad_feature = ",".join([str(v.entity_type["#PAD_TOKEN"])] * self.num_type_slots +
                                        [str(v.entity["#PAD_TOKEN"]), str(v.relation["#PAD_TOKEN"])])
        self.max_length = None
        self.missed = 0

    # ---- movie_data_format.py:81-99 ------------------------------------------------------------
    def scan_max_length(self, input_dir, files=TRAIN_FILES):
        max_length = -1
        for name in files:
            with open(os.path.join(input_dir, name)) as f:
                for line in f:
                    split = line.split("\t")
                    for path in split[2].strip().split("###"):
                        path_len = len(path.split("-"))
                        if not self.only_relation:
                            path_len = path_len // 2 + 2
                        if path_len > max_length:
                            max_length = path_len
        self.max_length = min(self.max_possible, max_length)
        return self.max_length

    # ---- :102-112 --------------------------------------------------------------------------------
    def _types_in_order(self, entity_types, length):
        assert length <= len(entity_types)
        ids = [self.v.entity_type[t] if t in self.v.entity_type else self.v.entity_type["#UNK_ENTITY_TYPE"] for t in entity_types]
        ids = sorted(ids)[:length][::-1]  # sorted as STRINGS (the vocab values are never converted), truncated, reversed
        return ",".join(str(i) for i in ids)

    # ---- :116-124 --------------------------------------------------------------------------------
    def _feature_only_relation(self, relation):
        v = self.v.relation
        return str(v[relation]) if relation in v else str(v["#UNK_RELATION"])

    # ---- :127-159 --------------------------------------------------------------------------------
    def _feature(self, prev_entity, relation):
        v = self.v
        out = ""
        if prev_entity in v.entity_type_map:
            types = v.entity_type_map[prev_entity]
            length = min(self.num_type_slots, len(types))
            for _ in range(self.num_type_slots - len(types)):
                out += str(v.entity_type["#PAD_TOKEN"]) + ","
            out += self._types_in_order(types, length) + ","
        else:
            for _ in range(self.num_type_slots):
                out += str(v.entity_type["#UNK_ENTITY_TYPE"]) + ","
        out += (str(v.entity[prev_entity]) if prev_entity in v.entity else str(v.entity["#UNK_ENTITY"])) + ","
        out += str(v.relation[relation]) if relation in v.relation else str(v.relation["#UNK_RELATION"])
        assert len(out.split(",")) == self.num_type_slots + 2
        return out

    def _padding(self, n):
        return " ".join([self.pad_feature] * n)

    # ---- one pair: :229-300 ----------------------------------------------------------------------
    def format_pair(self, e1, e2, paths_field):
        """-> the ';'-joined kept paths of the pair ('' when none survives)"""
        output_line = ""
        flag = 0
        for path_counter, each_path in enumerate(paths_field.split("###")):
            prev_entity = e1
            each_path = each_path.strip()
            tokens = each_path.split("-")
            path_len = len(tokens)
            if not self.only_relation:
                path_len = path_len // 2 + 2
            if path_len > self.max_length:
                continue
            num_pad = self.max_length - path_len
            if self.get_only_relation and not self.only_relation:
                num_pad += 1
            vec = self._padding(num_pad)
            for token_counter, token in enumerate(tokens):
                if not self.only_relation:
                    if token_counter % 2 == 0:  # relation, paired with the entity it leaves from
                        feat = self._feature_only_relation(token) if self.get_only_relation else self._feature(prev_entity, token)
                        vec = vec + feat if (token_counter == 0 and vec == "") else vec + " " + feat
                    else:
                        prev_entity = token
                else:
                    feat = self._feature_only_relation(token)
                    vec = vec + feat if (token_counter == 0 and vec == "") else vec + " " + feat
            if not self.only_relation and not self.get_only_relation:
                vec = vec + " " + self._feature(e2, "#END_RELATION")
            if len(vec.split(" ")) != self.max_length:  # the reference prints and skips (:285-295)
                continue
            if path_counter == 0 or flag == 0:
                flag = 1
                output_line += vec
            else:
                output_line += ";" + vec
        return output_line

    # ---- the file loop: :192-316 -----------------------------------------------------------------
    def run(self, input_dir, out_dir, files=TRAIN_FILES):
        if self.max_length is None:
            self.scan_max_length(input_dir, files)
        for d in ("train", "test"):
            p = os.path.join(out_dir, d)
            os.makedirs(p, exist_ok=True)
            for f in os.listdir(p):
                if os.path.exists(os.path.join(p, f)):
                    os.remove(os.path.join(p, f))
        self.missed = 0
        label = ""
        handles = {}
        try:
            for counter, name in enumerate(files):
                if counter in (0, 1):
                    output_file = os.path.join(out_dir, "train", "train.txt")
                    label = "1" if counter == 0 else "-1"
                else:
                    output_file = os.path.join(out_dir, "test", "test.txt")
                with open(os.path.join(input_dir, name)) as f:
                    for line in f:
                        split = line.split("\t")
                        if counter >= 2:
                            label = str(split[3].strip())
                        e1, e2 = split[0].strip(), split[1].strip()
                        output_line = self.format_pair(e1, e2, split[2])
                        path_count = len(output_line.split(";"))
                        int_label = str(self.v.label2int["domain"][label.strip()])
                        output_line = output_line.strip()
                        if len(output_line) == 0:
                            self.missed += 1
                            continue
                        output_line = (int_label + "\t" + output_line).strip()
                        fn = output_file + "." + str(path_count) + ".int"
                        if fn not in handles:
                            handles[fn] = open(fn, "a")
                        handles[fn].write(output_line + "\n")
        finally:
            for h in handles.values():
                h.close()
        return sorted(handles)


# ---- split_data.py:12-29 -----------------------------------------------------------------------
def split_data(file_name, num_lines, pre_str):
    """cut an .int file into `<pre_str>_part_<k>.int` pieces of num_lines lines (a trailing empty piece is
    created when the line count is a multiple of num_lines, as the reference does)"""
    count, line_count, out = 0, 0, []
    w = open(pre_str + "_part_%d.int" % count, "w", encoding="utf-8")
    out.append(w.name)
    with open(file_name, "r", encoding="utf-8") as r:
        for line in r:
            w.write(line)
            line_count += 1
            if line_count % num_lines == 0:
                w.close()
                count += 1
                w = open(pre_str + "_part_%d.int" % count, "w", encoding="utf-8")
                out.append(w.name)
    w.close()
    return out


# ---- movie_data_list.py:12-19 ------------------------------------------------------------------
def write_lists(data_dir):
    for type_str in ("train", "test"):
        path = os.path.join(data_dir, type_str)
        with open(os.path.join(data_dir, type_str + ".list"), "w", encoding="utf-8") as w:
            for file_name in os.listdir(path):
                if ".torch" in file_name:
                    w.write(type_str + "/" + file_name + "\n")


# ---- format_entity_pair.py:12-48 ---------------------------------------------------------------
def format_entity_pair(data_dir, list_name, pad_entity_id="2851219"):
    """<list>.entity: `label \\t first non-pad entity id \\t last entity id` of each pair's FIRST path"""
    with open(os.path.join(data_dir, list_name), "r", encoding="utf-8") as r:
        test_list = [line.strip() for line in r.readlines()]
    n = 0
    with open(os.path.join(data_dir, list_name + ".entity"), "w", encoding="utf-8") as w:
        for file_name in test_list:
            with open(os.path.join(data_dir, file_name.replace("torch", "int")), "r", encoding="utf-8") as ir:
                for line in ir:
                    parts = line.strip().split("\t")
                    items = parts[1].split(";")[0].split(" ")
                    start = ""
                    for item in items:
                        t = item.split(",")
                        if t[1] != pad_entity_id:
                            start = t[1]
                            break
                    w.write(parts[0] + "\t" + start + "\t" + items[-1].split(",")[1] + "\n")
                    n += 1
    return n


def main(argv=None):
    ap = argparse.ArgumentParser(description="movie_data_format.py, Python 3 (same flags)")
    ap.add_argument("-i", "--input_dir", required=True)
    ap.add_argument("-d", "--output_dir", required=True)
    ap.add_argument("-o", "--only_relation", required=True)
    ap.add_argument("-g", "--get_only_relation", required=True)
    ap.add_argument("-e", "--ec2_instance", required=True)
    ap.add_argument("-m", "--max_path_length", required=True)
    ap.add_argument("-t", "--max_num_types", required=True)
    ap.add_argument("--vocab_dir", default="vocab")
    a = ap.parse_args(argv)
    fmt = PathFormatter(Vocabs(a.vocab_dir), a.max_path_length, a.max_num_types, a.only_relation == "1", a.get_only_relation == "1")
    print("Max length is " + str(fmt.scan_max_length(a.input_dir)))
    print("pad features:", fmt.pad_feature)
    fmt.run(a.input_dir, a.output_dir)
    print("Missed entity pair count " + str(fmt.missed))


if __name__ == "__main__":
    main()
