"""The CLI's input-file form: `vadc FILE` starts ffmpeg and reads its stdout in place of stdin (vadc.c:531-608, :810-815, :1225-1229).
Here a stand-in named `ffmpeg` at the head of PATH records the argument vector it was started with (CPU: the child is started before the engine is created, so the
vector is there whether or not this machine has a GPU) and, on the GPU, plays a PCM file: the segments are the ones of the same audio on stdin."""
import os
import stat
import struct
import subprocess
import time

import numpy as np
import pytest

from conftest import ROOT, WEIGHTS

EXE = os.path.join(ROOT, "host", "vadc_hip")


def _exe():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "host"), "vadc_hip"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(EXE):
        pytest.skip("host/vadc_hip does not build here (the library is built by __graft_entry__.build()): " + r.stderr[-300:])
    return EXE


def _stand_in(tmp_path, pcm_file=None):
    """a program called ffmpeg: writes its arguments (one per line) and a line more if its stdin is not /dev/null, then plays `pcm_file` (if any) on stdout"""
    d = tmp_path / "bin"
    d.mkdir()
    log = tmp_path / "ffmpeg_args.txt"
    body = ["#!/bin/sh", f'for a in "$@"; do printf "%s\\n" "$a"; done > "{log}.tmp"',
            f'[ "$(readlink /proc/$$/fd/0)" = /dev/null ] || echo STDIN_IS_NOT_DEV_NULL >> "{log}.tmp"',
            f'mv "{log}.tmp" "{log}"']
    if pcm_file is not None:
        body.append(f'cat "{pcm_file}"')
    p = d / "ffmpeg"
    p.write_text("\n".join(body) + "\n")
    p.chmod(p.stat().st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    env = dict(os.environ, PATH=f"{d}:{os.environ.get('PATH', '')}")
    return env, log


def _wait_for(path, seconds=10.0):
    t0 = time.time()
    while not os.path.exists(path) and time.time() - t0 < seconds:
        time.sleep(0.05)
    assert os.path.exists(path), "the ffmpeg stand-in was not started"
    return open(path).read().splitlines()


def test_ffmpeg_argument_vector_is_the_references(tmp_path):
    """vadc.c:537: ffmpeg -hide_banner -loglevel error -nostats -ss %f -i "FILE" -map 0:a:%d -vn -sn -dn -ac 1 -ar 16k -f s16le -   (the file name as ONE argument,
    spaces and quotes included: no shell in between); the LAST bare argument is the file (vadc.c:1225-1229); our stdin is not handed on"""
    env, log = _stand_in(tmp_path)
    name = 'a clip "with" spaces.mkv'
    subprocess.run([_exe(), "--model", WEIGHTS, "first.wav", "--audio_source", "2", "--start_seconds", "12.5", name], input=b"\x01\x02" * 4096, env=env,
                   capture_output=True, timeout=120)
    got = _wait_for(log)
    assert got == ["-hide_banner", "-loglevel", "error", "-nostats", "-ss", "12.500000", "-i", name, "-map", "0:a:2", "-vn", "-sn", "-dn", "-ac", "1", "-ar", "16k",
                   "-f", "s16le", "-"]


def test_ffmpeg_defaults_and_the_8khz_container(tmp_path):
    """no --audio_source / --start_seconds: stream 0 from 0.000000 (vadc.c:1118-1119); a 37-tensor container (the v4 graph's 8 kHz branch) asks ffmpeg for 8 kHz"""
    env, log = _stand_in(tmp_path)
    subprocess.run([_exe(), "--model", WEIGHTS, "x.flac"], input=b"", env=env, capture_output=True, timeout=120)
    got = _wait_for(log)
    assert got[got.index("-ss") + 1] == "0.000000" and got[got.index("-map") + 1] == "0:a:0" and got[got.index("-ar") + 1] == "16k"
    os.remove(log)
    fake = tmp_path / "header_only.testtensor"
    fake.write_bytes(struct.pack("<ii", 1, 37))                 # the header is all that is read before the child starts; create then refuses the file
    r = subprocess.run([_exe(), "--model", str(fake), "x.flac"], input=b"", env=env, capture_output=True, timeout=120)
    got = _wait_for(log)
    assert got[got.index("-ar") + 1] == "8k"
    assert r.returncode != 0


def test_without_ffmpeg_on_the_path_the_cli_says_so(tmp_path):
    """vadc.c:568-572 "Error launching ffmpeg"; nothing is run"""
    empty = tmp_path / "empty"
    empty.mkdir()
    r = subprocess.run([_exe(), "--model", WEIGHTS, "clip.wav"], input=b"", env=dict(os.environ, PATH=str(empty)), capture_output=True, timeout=60)
    assert r.returncode != 0 and b"Error launching ffmpeg" in r.stderr and r.stdout == b""


@pytest.mark.gpu
def test_file_through_ffmpeg_gives_the_segments_of_the_same_audio_on_stdin(tmp_path):
    """the stand-in plays 8.2 windows of synthetic speech (one partial window, one partial chunk at the end): same `start,end` lines, same %f lines, as on stdin"""
    from vadc_amd import synth
    pcm = np.concatenate([synth.make_streams(1, 96 * 8 + 23, seed0=77)[0], np.zeros(500, np.int16)])
    f = tmp_path / "clip.s16le"
    pcm.tofile(f)
    env, log = _stand_in(tmp_path, pcm_file=f)
    for args in ((), ("--raw_probabilities",), ("--output_centi_seconds", "--min_silence", "100")):
        a = subprocess.run([_exe(), "--model", WEIGHTS, *args], input=pcm.tobytes(), capture_output=True, timeout=300)
        b = subprocess.run([_exe(), "--model", WEIGHTS, *args, "clip.mkv"], input=b"", env=env, capture_output=True, timeout=300)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr.decode(), b.stderr.decode())
        assert a.stdout == b.stdout and len(a.stdout) > 0
    assert "STDIN_IS_NOT_DEV_NULL" not in _wait_for(log)


@pytest.mark.gpu
def test_ffmpeg_that_fails_leaves_no_segments_and_a_message(tmp_path):
    d = tmp_path / "bin"
    d.mkdir()
    p = d / "ffmpeg"
    p.write_text("#!/bin/sh\necho 'clip.mkv: No such file or directory' >&2\nexit 1\n")
    p.chmod(0o755)
    r = subprocess.run([_exe(), "--model", WEIGHTS, "clip.mkv"], input=b"", env=dict(os.environ, PATH=f"{d}:{os.environ.get('PATH', '')}"), capture_output=True, timeout=300)
    assert r.stdout == b"" and b"No such file or directory" in r.stderr and b"ffmpeg gave no audio" in r.stderr
