"""Drop-in for GpsJammerApp/app/worker.py of mfkiwl/GPS-JAMMING, MI355X-backed.

``GPSAnalysisThread`` keeps the reference's constructor, its seven Qt signals, the
attributes and methods the main window touches (GpsJammerApp/app/ui_mainwindow.py:684-698,
713-735,746-826,853-900) and the result formats (worker.py:549-565, 590-606).

What changed underneath:
* ``precalculate_power_profile`` (worker.py:198-275) -- the Python loop over 65 536-byte
  chunks is ONE call into libgpsjam_hip.so (kernel K1: packed-byte dot products, exact
  integer chunk sums) on the memory-mapped capture; the 5th-percentile floor, the +6 dB
  threshold and the byte ranges are derived from the returned float32 power map with the
  same numpy expressions, so ``power_map`` / ``global_baseline_power`` /
  ``jamming_byte_ranges`` have the reference's types;
* the triangulation goes through this package's ``triangulateRSSI`` (K3 on the GPU).
The telemetry state machine (worker.py:277-475) and the gnssdec orchestration (:477-565) are
host-side control flow and behave as in the reference.  There is no CPU fallback for the
scans: without the HIP library the scan reports the error and ``power_map_ready`` stays False.
"""
import json
import os
import subprocess
import sys
import threading
from collections import deque
from http.server import BaseHTTPRequestHandler, HTTPServer

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (_PKG_ROOT, os.path.join(_PKG_ROOT, "skrypty")):
    if _p not in sys.path:
        sys.path.append(_p)

import gpsjam                                                   # noqa: E402
from gpsjam.qt_compat import QThread, Signal                    # noqa: E402
from triangulateRSSI import triangulate_jammer_location         # noqa: E402


class ReusableHTTPServer(HTTPServer):
    allow_reuse_address = True


class _DataReceiverHandler(BaseHTTPRequestHandler):
    """Receives gnssdec's JSON POSTs on /data (GpsJammerApp/backend/sdrout.c:10-57)."""
    thread_instance = None

    def _reply(self, code, body=None):
        try:
            self.send_response(code)
            if body is not None:
                self.send_header('Content-Type', 'application/json')
                self.send_header('Content-Length', str(len(body)))
            self.end_headers()
            if body is not None:
                self.wfile.write(body)
        except (BrokenPipeError, ConnectionResetError):
            pass

    def do_POST(self):
        if self.path != '/data':
            self._reply(404)
            return
        try:
            size = int(self.headers.get('Content-Length', 0))
            record = json.loads(self.rfile.read(size).decode('utf-8'))
        except json.JSONDecodeError:
            print("Błąd parsowania JSON")
            self._reply(400)
            return
        except Exception as e:
            print(f"[HTTP HANDLER] Błąd: {e}")
            self._reply(500)
            return
        try:
            if self.thread_instance:
                self.thread_instance.process_incoming_data(record)
            self._reply(200, b'{"status":"ok"}')
        except Exception as e:
            print(f"[HTTP HANDLER] Błąd: {e}")
            self._reply(500)

    def log_message(self, format, *args):
        pass


REASON_POWER = "Moc (Mapowana)"
REASON_QUALITY = "Jakość/Integrity"


class GPSAnalysisThread(QThread):

    analysis_complete = Signal(list)
    progress_update = Signal(int, str)
    new_analysis_text = Signal(str)
    new_position_data = Signal(float, float, float)
    jamming_analysis_complete = Signal(list)
    triangulation_complete = Signal(dict)
    jamming_detected_realtime = Signal(bool, dict)

    def __init__(self, file_paths, power_threshold=6.0, antenna_positions=None, satellite_system='GPS',
                 hold_position=False):
        super().__init__()
        self.file_paths = file_paths
        self.POWER_CHUNK_SIZE = 32768
        # the reference ignores power_threshold in favour of a fixed +6 dB (worker.py:85-86)
        self.THRESHOLD_POWER_RISE_DB = 6.0
        # what this thread's own triangulation passes to triangulate_jammer_location (the reference: worker.py:598);
        # the power scan asks for the amplitude statistics at THIS threshold while the first file uploads
        self.TRIANGULATION_RSSI_THRESHOLD = 0.0
        print(f"[GPS THREAD] Próg detekcji mocy (ITU-R): {self.THRESHOLD_POWER_RISE_DB} dB")
        self.THRESHOLD_CN0_DROP_DB = 8.0
        self.THRESHOLD_RESIDUALS_MEDIAN_M = 40.0
        self.THRESHOLD_RESIDUAL_SINGLE_SAT_M = 800.0
        self.MIN_BAD_SATS_FOR_ALARM = 2
        self.THRESHOLD_HGT_MAX = 10000.0
        self.THRESHOLD_GDOP_MAX = 6.0
        self.THRESHOLD_NSAT_MIN = 4

        self.antenna_positions = antenna_positions if antenna_positions else {
            'antenna1': [0.0, 0.0], 'antenna2': [0.5, 0.0], 'antenna3': [0.0, 0.5]}
        self.satellite_system = satellite_system
        self.gnss_system_flag = {'GPS': '-g', 'GLONASS': '-n', 'Galileo': '-l'}.get(satellite_system, '-g')
        self.hold_position = hold_position

        # telemetry state
        self.current_buffcnt = 0
        self.current_lat = self.current_lon = self.current_hgt = 0.0
        self.current_nsat = 0
        self.current_gdop = self.current_clk_bias = self.current_signal_time = 0.0
        self.current_cn0_avg = 0.0
        self.current_residuals_median = 0.0
        self.current_residuals_bad_count = 0

        # power scan state
        self.power_map = []
        self.global_baseline_power = 0.0
        self.current_iq_power = 0.0
        self.power_map_ready = False
        self.total_file_bytes = 0
        self.power_detection_enabled = True
        self.jamming_start_byte_offset = None
        self.jamming_byte_ranges = []          # the reference only creates it inside the scan

        self.cn0_history = deque(maxlen=100)
        self.median_cn0 = 0.0

        # detector state
        self.jamming_detected = False
        self.jamming_events = []
        self.potential_jamming_start_signal_time = None
        self.potential_jamming_end_signal_time = None
        self.potential_start_buffcnt = 0
        self.active_event_start_buffcnt = 0
        self.active_event_start_time = 0.0
        self.required_jamming_duration_sec = 2.5
        self.required_clean_duration_sec = 2.0
        self.last_safe_position_buffer = deque(maxlen=50)
        self.last_position_before_jamming = {'lat': 0.0, 'lon': 0.0, 'hgt': 0.0, 'buffcnt': 0, 'valid': False}

        self.http_server = None
        self.http_thread = None
        self.jamming_thread = None
        self.triangulation_thread = None
        self.total_samples = 0
        self.estimated_total_samples = 0
        self.triangulation_result = None
        self.stop_requested = False
        self.triangulation_started = False
        self.scan_kernel_ms = 0.0

        self.triangulation_complete.connect(self.on_triangulation_complete)
        self.gnssdec_path = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                         "..", "backend", "bin", "gnssdec"))
        if self.file_paths:
            self.calculate_file_samples()

    # ------------------------------------------------------------------ power scan
    def calculate_file_samples(self):
        try:
            if not self.file_paths or not os.path.exists(self.file_paths[0]):
                return
            self.total_file_bytes = os.path.getsize(self.file_paths[0])
            self.total_samples = self.total_file_bytes // 2          # 2 bytes per I/Q pair
            self.estimated_total_samples = self.total_samples
        except Exception as e:
            print(f"[PROGRESS] Błąd przy obliczaniu próbek: {e}")
            self.total_samples = 0

    def precalculate_power_profile(self):
        """Power map of the first capture + the F1 byte ranges, on the GPU."""
        if not self.file_paths or not os.path.exists(self.file_paths[0]):
            return
        path = self.file_paths[0]
        print(f"[POWER SCAN] Skanowanie (uint8): {os.path.basename(path)}")
        self.progress_update.emit(0, "scanning_power")
        try:
            chunk_bytes = self.POWER_CHUNK_SIZE * 2
            self.total_file_bytes = os.path.getsize(path)
            if self.stop_requested:
                self.power_map = np.array([])
            else:
                # one upload; the power scan runs on the pieces of the file while the rest is still uploading, and
                # the amplitude statistics THIS thread's triangulation will ask for -- it calls
                # triangulate_jammer_location(threshold=TRIANGULATION_RSSI_THRESHOLD), as the reference does
                # (worker.py:598: threshold=0.0) -- come out of the same pass over the bytes and ride on the capture
                dev = gpsjam.default_device()
                dev.last_kernel_ms = 0.0
                cap = gpsjam.resident_capture(path, chunk_bytes=chunk_bytes, eps=1e-10,
                                              rssi_threshold=self.TRIANGULATION_RSSI_THRESHOLD)
                self.power_map = dev.chunk_power(cap, chunk_bytes=chunk_bytes, eps=1e-10)
                self.scan_kernel_ms = dev.last_kernel_ms
                if self.scan_kernel_ms > 0:
                    rate = self.total_file_bytes / 2 / self.scan_kernel_ms / 1e3
                    print(f"[POWER SCAN] GPU: {self.scan_kernel_ms:.3f} ms, {rate:.0f} Msamples/s, "
                          f"{self.total_file_bytes / self.scan_kernel_ms / 1e6:.0f} GB/s")
            # the reference emits 0..10 % in even steps while it loops (worker.py:234-237)
            for pct in (2, 4, 6, 8, 10):
                self.progress_update.emit(pct, "scanning_power")

            if len(self.power_map) > 0:
                self.global_baseline_power = np.percentile(self.power_map, 5)
                if self.global_baseline_power <= 0:
                    self.global_baseline_power = 1.0
                limit = self.global_baseline_power * 10 ** (self.THRESHOLD_POWER_RISE_DB / 10.0)
                hot = self.power_map > limit
                self.jamming_byte_ranges = []
                if hot.any():
                    step = np.diff(hot.astype(int))
                    begins = np.where(step == 1)[0] + 1
                    ends = np.where(step == -1)[0] + 1
                    if hot[0]:
                        begins = np.insert(begins, 0, 0)
                    if hot[-1]:
                        ends = np.append(ends, len(self.power_map))
                    self.jamming_byte_ranges = [(b * chunk_bytes, e * chunk_bytes) for b, e in zip(begins, ends)]
                    print(f"[POWER SCAN] Wykryto {len(self.jamming_byte_ranges)} okresów wysokiej mocy (F1).")
                else:
                    print(f"[POWER SCAN] Nie wykryto skoku mocy powyżej progu {self.THRESHOLD_POWER_RISE_DB} dB.")
            self.power_map_ready = True
            self.progress_update.emit(10, "scanning_power_done")
        except Exception as e:
            print(f"[POWER SCAN] BŁĄD: {e}")
            self.power_map_ready = False

    # ------------------------------------------------------------------ telemetry
    def _absorb_position(self, position, observations):
        self.current_buffcnt = position.get('buffcnt', 0)
        self.current_lat = float(position.get('lat', 0.0))
        self.current_lon = float(position.get('lon', 0.0))
        self.current_hgt = float(position.get('hgt', 0.0))
        self.current_nsat = position.get('nsat', 0)
        self.current_gdop = float(position.get('gdop', 0.0))
        self.current_clk_bias = float(position.get('clk_bias', 0.0))

        if self.power_map_ready and self.total_file_bytes > 0:
            frac = min(1.0, max(0.0, self.current_buffcnt / self.total_file_bytes))
            slot = min(int(frac * len(self.power_map)), len(self.power_map) - 1)
            self.current_iq_power = self.power_map[slot]

        snr = [o.get('snr', 0.0) for o in observations if 'snr' in o]
        self.current_cn0_avg = 0.0
        self.current_residuals_median = 0.0
        self.current_residuals_bad_count = 0
        if snr:
            self.current_cn0_avg = np.mean(snr)
            resid = [o.get('residual', 0.0) for o in observations if 'residual' in o]
            if resid:
                self.current_residuals_median = np.median(resid)
                self.current_residuals_bad_count = sum(1 for r in resid if r > self.THRESHOLD_RESIDUAL_SINGLE_SAT_M)

        if not self.jamming_detected and self.current_cn0_avg > 0:
            self.cn0_history.append(self.current_cn0_avg)
        self.median_cn0 = np.median(self.cn0_history) if len(self.cn0_history) > 10 else self.current_cn0_avg

        self.update_progress_bar()

        clean = not self.jamming_detected
        if self.jamming_byte_ranges and self.current_buffcnt >= self.jamming_byte_ranges[0][0]:
            clean = False
        if clean and self.current_lat != 0.0 and self.current_nsat >= 4:
            self.last_position_before_jamming = {'lat': self.current_lat, 'lon': self.current_lon,
                                                 'hgt': self.current_hgt, 'buffcnt': self.current_buffcnt,
                                                 'valid': True}
        self.check_jamming_conditions()

    def process_incoming_data(self, data):
        """One gnssdec telemetry record (worker.py:277-361)."""
        try:
            position = data.get('position', {})
            observations = data.get('observations', [])
            try:
                self.current_signal_time = float(data.get('elapsed_time', 0.0))
            except Exception:
                pass
            if position:
                self._absorb_position(position, observations)

            rise_db = 0.0
            if self.global_baseline_power > 0 and self.current_iq_power > 0:
                rise_db = 10 * np.log10(self.current_iq_power / self.global_baseline_power)
            self.new_analysis_text.emit(f"[{self.current_signal_time:.2f}s, Lat:{self.current_lat:.6f}, "
                                        f"Lon:{self.current_lon}, Pwr:{rise_db:.1f}dB]")
            if not self.jamming_detected and (self.current_lat != 0.0 or self.current_lon != 0.0):
                self.new_position_data.emit(self.current_lat, self.current_lon, self.current_hgt)
        except Exception as e:
            print(f"[WORKER] Błąd: {e}")

    def _in_power_range(self):
        for lo, hi in self.jamming_byte_ranges or []:
            if lo <= self.current_buffcnt <= hi:
                return lo
        return None

    def check_jamming_conditions(self):
        """F1 power map / F2 C/N0 drop / altitude sanity -> start and end of events
        (worker.py:363-413).  As in the reference, the residual integrity test is computed
        but does not feed the decision (its flag is never set, :379-388)."""
        f1 = self._in_power_range() is not None
        f2 = len(self.cn0_history) > 40 and self.current_cn0_avg < (self.median_cn0 - self.THRESHOLD_CN0_DROP_DB)
        altitude_off = self.current_nsat > 0 and abs(self.current_hgt) > self.THRESHOLD_HGT_MAX
        jamming_now = f1 or f2 or (altitude_off and self.current_nsat > 0)

        if not self.jamming_detected:
            if not jamming_now:
                self.potential_jamming_start_signal_time = None
            elif f1:
                self.confirm_jamming_start(reason=REASON_POWER)
            elif self.potential_jamming_start_signal_time is None:
                self.potential_jamming_start_signal_time = self.current_signal_time
                self.potential_start_buffcnt = self.current_buffcnt
            elif (self.current_signal_time - self.potential_jamming_start_signal_time) >= self.required_jamming_duration_sec:
                self.confirm_jamming_start(reason=REASON_QUALITY)
            return

        if jamming_now:
            self.potential_jamming_end_signal_time = None
        elif self.potential_jamming_end_signal_time is None:
            self.potential_jamming_end_signal_time = self.current_signal_time
        elif (self.current_signal_time - self.potential_jamming_end_signal_time) >= self.required_clean_duration_sec:
            self.confirm_jamming_end()
            self.potential_jamming_end_signal_time = None

    def confirm_jamming_start(self, reason="N/A"):
        self.jamming_detected = True
        start_byte = self.potential_start_buffcnt
        if reason == REASON_POWER and self.jamming_byte_ranges:
            lo = self._in_power_range()
            if lo is not None:
                start_byte = lo
        else:
            start_byte = self.potential_start_buffcnt if self.potential_start_buffcnt > 0 else self.current_buffcnt
        self.active_event_start_buffcnt = start_byte
        self.active_event_start_time = self.current_signal_time
        if reason == REASON_QUALITY and self.potential_jamming_start_signal_time:
            self.active_event_start_time = self.potential_jamming_start_signal_time
        print(f"[DETEKTOR] 🚨 ATAK POTWIERDZONY! Powód: {reason}")
        print(f"[DETEKTOR]    Start: {self.active_event_start_time:.2f}s")
        if self.last_position_before_jamming['valid']:
            self.jamming_detected_realtime.emit(True, self.last_position_before_jamming)
        else:
            print("[DETEKTOR] ⚠️ Brak bezpiecznej pozycji przed atakiem!")

    def _close_event(self):
        end_time = self.current_signal_time
        self.jamming_events.append({
            'start_sample': self.active_event_start_buffcnt,
            'end_sample': self.current_buffcnt,
            'start_time': self.active_event_start_time,
            'end_time': end_time,
            'duration': end_time - self.active_event_start_time,
        })
        return end_time

    def confirm_jamming_end(self):
        self.jamming_detected = False
        end_time = self._close_event()
        print(f"[DETEKTOR] ✅ Koniec ataku. (Koniec: {end_time:.2f}s)")
        self.jamming_detected_realtime.emit(False, {})

    def get_best_safe_position(self):
        if not self.last_safe_position_buffer:
            return self.last_position_before_jamming
        return self.last_safe_position_buffer[0]

    def update_progress_bar(self):
        if self.total_samples > 0 and self.current_buffcnt > 0:
            total = max(self.total_samples, self.estimated_total_samples)
            pct = min(100, int((self.current_buffcnt / total) * 100))
            if not self.jamming_detected:
                state = "normal"
            elif self.triangulation_thread and self.triangulation_thread.is_alive():
                state = "triangulating"
            else:
                state = "jamming"
            self.progress_update.emit(pct, state)
        else:
            self.progress_update.emit(0, "normal")

    # ------------------------------------------------------------------ orchestration
    def run(self):
        self.stop_requested = False
        _DataReceiverHandler.thread_instance = self
        print("[WORKER] Uruchamianie wątku analizy...")
        try:
            self.http_server = ReusableHTTPServer(('127.0.0.1', 1234), _DataReceiverHandler)
            self.http_thread = threading.Thread(target=self.http_server.serve_forever)
            self.http_thread.daemon = True
            self.http_thread.start()
            print("[WORKER] Serwer HTTP uruchomiony (port 1234).")
        except Exception as e:
            print(f"[WORKER] Błąd serwera: {e}")
            return

        first = self.file_paths[0] if self.file_paths else None
        if not first:
            return
        self.precalculate_power_profile()
        if self.stop_requested:
            self.shutdown_server()
            return

        try:
            print(f"[WORKER] Uruchamianie analizy {self.gnssdec_path}...")
            command = [self.gnssdec_path, self.gnss_system_flag]
            if self.hold_position:
                command.append('-h')
            command.append(first)
            self.current_signal_time = 0.0
            self.cn0_history.clear()
            subprocess.run(command, check=True, capture_output=True, text=True)
            print("[WORKER] Analiza gnssdec zakończona.")
        except Exception as e:
            print(f"[WORKER] Błąd procesu gnssdec: {e}")
        finally:
            self.progress_update.emit(100, "completed")
            self.shutdown_server()
            if self.jamming_detected:
                # the file ended while an event was open: close it and triangulate (:528-547)
                print("[WORKER] Plik zakończony w trakcie aktywnego jammingu. Zamykanie zdarzenia.")
                self._close_event()
                print("[WORKER] Uruchamiam triangulację na koniec pliku...")
                self.analyze_triangulation_after_gnssdec()
                if self.triangulation_thread:
                    self.triangulation_thread.join()
            self.analysis_complete.emit(self.build_result_list())

    def build_result_list(self):
        """The list handed to analysis_complete (worker.py:549-565)."""
        if not self.jamming_events:
            return [{'type': 'no_jamming'}]
        return [{'type': 'jamming', 'event_number': n, 'start_sample': ev['start_sample'],
                 'end_sample': ev['end_sample'], 'start_time': ev['start_time'], 'end_time': ev['end_time'],
                 'duration': ev['duration'], 'triangulation': self.triangulation_result}
                for n, ev in enumerate(self.jamming_events, 1)]

    def analyze_triangulation_after_gnssdec(self):
        def job():
            try:
                if len(self.file_paths) < 2:
                    print("[TRIANGULACJA] Za mało plików do triangulacji.")
                    return
                print("[TRIANGULACJA] Start obliczeń...")
                anchor = self.last_position_before_jamming
                if anchor['valid']:
                    ref_lat, ref_lon = anchor['lat'], anchor['lon']
                else:
                    ref_lat = self.current_lat if self.current_lat != 0.0 else 50.0
                    ref_lon = self.current_lon if self.current_lon != 0.0 else 20.0
                result = triangulate_jammer_location(
                    file_paths=self.get_test_files_for_triangulation(),
                    antenna_positions_meters=[np.array(self.antenna_positions[k])
                                              for k in ('antenna1', 'antenna2', 'antenna3')],
                    reference_lat=ref_lat, reference_lon=ref_lon, tx_power=40.0, path_loss_exp=3.0,
                    frequency_mhz=1575.42, threshold=self.TRIANGULATION_RSSI_THRESHOLD, verbose=False)
                if result['success'] and anchor['valid']:
                    result['reference_position'] = anchor
                self.triangulation_result = result
                self.triangulation_complete.emit(result)
            except Exception as e:
                print(f"[TRIANGULACJA] Błąd: {e}")

        self.triangulation_thread = threading.Thread(target=job)
        self.triangulation_thread.start()

    def get_test_files_for_triangulation(self):
        """test<k>.bin next to the first capture when present, else the capture itself
        (worker.py:613-627)."""
        folder = os.path.dirname(self.file_paths[0]) if self.file_paths else "../data"
        chosen = []
        for k in range(min(len(self.file_paths), 3)):
            candidate = os.path.join(folder, f"test{k + 1}.bin")
            if os.path.exists(candidate):
                chosen.append(candidate)
            elif k < len(self.file_paths):
                chosen.append(self.file_paths[k])
        return chosen if chosen else self.file_paths

    def on_triangulation_complete(self, result):
        self.triangulation_result = result
        if result['success']:
            geo = result['location_geographic']
            print(f"[TRIANGULATION] 🎯 Wynik: {geo['lat']:.8f}, {geo['lon']:.8f}")

    def get_triangulation_result(self):
        return self.triangulation_result

    def on_jamming_detected(self, events):
        pass

    def shutdown_server(self):
        server, self.http_server = self.http_server, None
        if server:
            server.shutdown()
            if self.http_thread:
                self.http_thread.join()
            server.server_close()

    def get_current_position_data(self):
        return {'buffcnt': self.current_buffcnt, 'lat': self.current_lat, 'lon': self.current_lon,
                'nsat': self.current_nsat, 'jamming': self.jamming_detected}
