{-# LANGUAGE ForeignFunctionInterface #-}

{-| GPU back-end for the FM hot path of the @sdr@ library, over the C ABI of @libsdr_hip.so@
    (@include/sdr_hip.h@, layer 3: Pipe operators).

    NOT COMPILED ANYWHERE IN THIS REPOSITORY'S BUILD IMAGE (no GHC there): this is the binding a
    maintainer of adamwalker/sdr would add next to SDR.Filter / SDR.Demod / SDR.Util.  The tested
    boundary is the C ABI itself (tests/test_abi.py, tests/test_gpu_pipes.py).

    The operators mirror the reference's:

    > firDecimatorGpu :: GpuDecimator -> Int -> Pipe (Vector (Complex Float)) (Vector (Complex Float)) IO ()   -- firDecimator, Filter.hs:574
    > firResamplerGpu :: GpuResampler -> Int -> Pipe (Vector Float) (Vector Float) IO ()                       -- firResampler, Filter.hs:679
    > firFilterGpu    :: GpuFilter    -> Int -> Pipe (Vector Float) (Vector Float) IO ()                       -- firFilter,    Filter.hs:532
    > fmDemodGpu      :: Pipe (Vector (Complex Float)) (Vector Float) IO ()                                    -- fmDemod,      Demod.hs:40
    > dcBlockingFilterGpu :: Pipe (Vector Float) (Vector Float) IO ()                                          -- dcBlockingFilter, Filter.hs:730
    > interleavedIQUnsignedByteToFloatGpu :: Vector CUChar -> Vector (Complex Float)                           -- Util.hs:137
    > fmReceiverGpu   :: GpuFmChain -> Int -> Int -> Pipe (Vector CUChar) (Vector Float) IO ()                 -- fm.hs:34-41 as one operator

    and produce the same vectors, bit for bit, as the AVX variants the reference selects on any AVX
    host (the One/Cross split at buffer seams included).
-}
module SDR.GPU (
    GpuDecimator, GpuResampler, GpuFilter,
    gpuDecimatorC, gpuResamplerR, gpuFilterSymR, gpuFilterR,
    firDecimatorGpu, firResamplerGpu, firFilterGpu, fmDemodGpu, dcBlockingFilterGpu,
    interleavedIQUnsignedByteToFloatGpu,
    GpuFmChain, gpuFmChain, fmReceiverGpu
    ) where

import           Control.Monad
import           Data.Complex
import           Foreign
import           Foreign.C.String
import           Foreign.C.Types
import qualified Data.Vector.Storable         as VS
import           Pipes
import           System.IO.Unsafe             (unsafePerformIO)

data SdrDecimator
data SdrResampler
data SdrFilter
data SdrPipe
data SdrChain
data SdrStream

newtype GpuDecimator = GpuDecimator (Ptr SdrDecimator)
newtype GpuResampler = GpuResampler (Ptr SdrResampler)
newtype GpuFilter    = GpuFilter    (Ptr SdrFilter)
newtype GpuFmChain   = GpuFmChain   (Ptr SdrChain)

-- The imports are `safe`: the calls block on the device and must not stall the RTS.
foreign import ccall safe "sdrhip_last_error"          c_last_error        :: IO CString
foreign import ccall safe "sdrhip_decimator_create"    c_decimator_create  :: Ptr (Ptr SdrDecimator) -> CInt -> CInt -> CInt -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_resampler_create"    c_resampler_create  :: Ptr (Ptr SdrResampler) -> CInt -> CInt -> CInt -> CInt -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_filter_create"       c_filter_create     :: Ptr (Ptr SdrFilter) -> CInt -> CInt -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_filter_sym_create"   c_filter_sym_create :: Ptr (Ptr SdrFilter) -> CInt -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_fir_decimator"  c_pipe_decimator    :: Ptr (Ptr SdrPipe) -> Ptr SdrDecimator -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_fir_resampler"  c_pipe_resampler    :: Ptr (Ptr SdrPipe) -> Ptr SdrResampler -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_fir_filter"     c_pipe_filter       :: Ptr (Ptr SdrPipe) -> Ptr SdrFilter -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_fm_demod"       c_pipe_demod        :: Ptr (Ptr SdrPipe) -> IO CInt
foreign import ccall safe "sdrhip_pipe_push"           c_pipe_push         :: Ptr SdrPipe -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_pop"            c_pipe_pop          :: Ptr SdrPipe -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "sdrhip_pipe_dc_blocker"     c_pipe_dc_blocker   :: Ptr (Ptr SdrPipe) -> IO CInt
-- throughput knob: stage equal-sized pushes and submit them n at a time (same output blocks, n blocks of latency)
foreign import ccall safe "sdrhip_pipe_set_coalesce"   c_pipe_set_coalesce :: Ptr SdrPipe -> CInt -> IO CInt
foreign import ccall safe "sdrhip_fm_stream_set_coalesce" c_stream_set_coalesce :: Ptr SdrStream -> CInt -> IO CInt
foreign import ccall safe "sdrhip_fm_chain_create"     c_chain_create      :: Ptr (Ptr SdrChain) -> CInt -> CInt -> Ptr CFloat -> CInt -> CInt -> CInt -> Ptr CFloat -> CInt -> Ptr CFloat -> CInt -> CFloat -> Int64 -> IO CInt
foreign import ccall safe "sdrhip_fm_stream_create"    c_stream_create     :: Ptr (Ptr SdrStream) -> Ptr SdrChain -> CInt -> CInt -> IO CInt
foreign import ccall safe "sdrhip_fm_stream_push"      c_stream_push       :: Ptr SdrStream -> Ptr CUChar -> CInt -> IO CInt
foreign import ccall safe "sdrhip_fm_stream_pop"       c_stream_pop        :: Ptr SdrStream -> Ptr CFloat -> CInt -> IO CInt
foreign import ccall safe "convertCAVX"                c_convertCAVX       :: CInt -> Ptr CUChar -> Ptr CFloat -> IO ()

-- | SDRHIP_ORDER_AVX: reproduce the variant 'SDR.CPUID.featureSelect' picks on any AVX host.
orderAVX :: CInt
orderAVX = 2

-- | The reference's hand-rolled @assert@ calls @error@ (Filter.hs:526-527); so does a negative status.
check :: CInt -> IO CInt
check rc
    | rc < 0    = c_last_error >>= peekCString >>= \msg -> error ("sdr_hip: " ++ msg)
    | otherwise = return rc

withCoeffs :: [Float] -> (Ptr CFloat -> CInt -> IO a) -> IO a
withCoeffs cs act = withArrayLen (map realToFrac cs) $ \n p -> act p (fromIntegral n)

-- | 'fastDecimatorC' (Filter.hs:352-356) on the GPU: complex data, real taps.
gpuDecimatorC :: Int -> [Float] -> IO GpuDecimator
gpuDecimatorC factor coeffs = alloca $ \pp -> do
    _ <- withCoeffs coeffs $ \p n -> c_decimator_create pp orderAVX 1 (fromIntegral factor) p n >>= check
    GpuDecimator <$> peek pp

-- | 'fastResamplerR' (Filter.hs:468-473) on the GPU.
gpuResamplerR :: Int -> Int -> [Float] -> IO GpuResampler
gpuResamplerR interp decim coeffs = alloca $ \pp -> do
    _ <- withCoeffs coeffs $ \p n -> c_resampler_create pp orderAVX 0 (fromIntegral interp) (fromIntegral decim) p n >>= check
    GpuResampler <$> peek pp

-- | 'fastFilterSymR' (Filter.hs:258-261) on the GPU: pass the FIRST HALF of an even-length linear-phase filter.
gpuFilterSymR :: [Float] -> IO GpuFilter
gpuFilterSymR half = alloca $ \pp -> do
    _ <- withCoeffs half $ \p n -> c_filter_sym_create pp orderAVX p n >>= check
    GpuFilter <$> peek pp

-- | 'fastFilterR' (Filter.hs:191-194) on the GPU.
gpuFilterR :: [Float] -> IO GpuFilter
gpuFilterR coeffs = alloca $ \pp -> do
    _ <- withCoeffs coeffs $ \p n -> c_filter_create pp orderAVX 0 p n >>= check
    GpuFilter <$> peek pp

-- | Forward blocks through one C pipe.  @wIn@ / @wOut@: floats per element (2 for complex).
--   Output blocks have exactly @blockSizeOut@ elements (advanceOutBuf, Filter.hs:516-523).
runPipe :: (Storable a, Storable b) => Int -> Int -> Ptr SdrPipe -> Int -> Pipe (VS.Vector a) (VS.Vector b) IO ()
runPipe wIn wOut pipe blockSizeOut = forever $ do
    inp   <- await
    ready <- lift $ VS.unsafeWith (VS.unsafeCast inp) $ \ptr ->
                 c_pipe_push pipe ptr (fromIntegral (VS.length inp)) >>= check
    replicateM_ (fromIntegral ready) $ do
        out <- lift $ do
            fp <- mallocForeignPtrArray (wOut * blockSizeOut) :: IO (ForeignPtr CFloat)
            _  <- withForeignPtr fp $ \o -> c_pipe_pop pipe o (fromIntegral blockSizeOut) >>= check
            return $ VS.unsafeCast $ VS.unsafeFromForeignPtr0 fp (wOut * blockSizeOut)
        yield out
  where _ = wIn

mkPipe :: (Ptr (Ptr SdrPipe) -> IO CInt) -> IO (Ptr SdrPipe)
mkPipe create = alloca $ \pp -> create pp >>= check >> peek pp

firDecimatorGpu :: GpuDecimator -> Int -> Pipe (VS.Vector (Complex Float)) (VS.Vector (Complex Float)) IO ()
firDecimatorGpu (GpuDecimator d) blockSizeOut = do
    pipe <- lift $ mkPipe $ \pp -> c_pipe_decimator pp d (fromIntegral blockSizeOut)
    runPipe 2 2 pipe blockSizeOut

firResamplerGpu :: GpuResampler -> Int -> Pipe (VS.Vector Float) (VS.Vector Float) IO ()
firResamplerGpu (GpuResampler r) blockSizeOut = do
    pipe <- lift $ mkPipe $ \pp -> c_pipe_resampler pp r (fromIntegral blockSizeOut)
    runPipe 1 1 pipe blockSizeOut

firFilterGpu :: GpuFilter -> Int -> Pipe (VS.Vector Float) (VS.Vector Float) IO ()
firFilterGpu (GpuFilter f) blockSizeOut = do
    pipe <- lift $ mkPipe $ \pp -> c_pipe_filter pp f (fromIntegral blockSizeOut)
    runPipe 1 1 pipe blockSizeOut

-- | 'fmDemod' (Demod.hs:40-46): one output vector per input vector; the block size passed to
--   'runPipe' is only an upper bound here, the C side returns each block's own length.
fmDemodGpu :: Pipe (VS.Vector (Complex Float)) (VS.Vector Float) IO ()
fmDemodGpu = do
    pipe <- lift $ mkPipe c_pipe_demod
    forever $ do
        inp   <- await
        ready <- lift $ VS.unsafeWith (VS.unsafeCast inp) $ \ptr ->
                     c_pipe_push pipe ptr (fromIntegral (VS.length inp)) >>= check
        replicateM_ (fromIntegral ready) $ do
            out <- lift $ do
                let cap = VS.length inp
                fp  <- mallocForeignPtrArray cap :: IO (ForeignPtr CFloat)
                n   <- withForeignPtr fp $ \o -> c_pipe_pop pipe o (fromIntegral cap) >>= check
                return $ VS.unsafeCast $ VS.unsafeFromForeignPtr0 fp (fromIntegral n)
            yield out

-- | 'interleavedIQUnsignedByteToFloatFast' (Util.hs:137-138) through the drop-in symbol.
interleavedIQUnsignedByteToFloatGpu :: VS.Vector CUChar -> VS.Vector (Complex Float)
interleavedIQUnsignedByteToFloatGpu inBuf = unsafePerformIO $ do
    fp <- mallocForeignPtrArray (VS.length inBuf) :: IO (ForeignPtr CFloat)
    VS.unsafeWith inBuf $ \iPtr -> withForeignPtr fp $ \oPtr ->
        c_convertCAVX (fromIntegral $ VS.length inBuf) iPtr oPtr
    return $ VS.unsafeCast $ VS.unsafeFromForeignPtr0 fp (VS.length inBuf)

-- | 'dcBlockingFilter' (Filter.hs:730-739): one output vector per input vector, the filter state
--   carried on the device.  Worth it for long vectors only (a short block is one dependent chain).
dcBlockingFilterGpu :: Pipe (VS.Vector Float) (VS.Vector Float) IO ()
dcBlockingFilterGpu = do
    pipe <- lift $ mkPipe c_pipe_dc_blocker
    forever $ do
        inp   <- await
        ready <- lift $ VS.unsafeWith (VS.unsafeCast inp) $ \ptr ->
                     c_pipe_push pipe ptr (fromIntegral (VS.length inp)) >>= check
        replicateM_ (fromIntegral ready) $ do
            out <- lift $ do
                let cap = VS.length inp
                fp  <- mallocForeignPtrArray cap :: IO (ForeignPtr CFloat)
                n   <- withForeignPtr fp $ \o -> c_pipe_pop pipe o (fromIntegral cap) >>= check
                return $ VS.unsafeCast $ VS.unsafeFromForeignPtr0 fp (fromIntegral n)
            yield out

-- | The whole receiver of examples/fm/fm.hs:34-41 -- convert, decimate, demodulate, resample, filter,
--   gain -- with every intermediate resident in device memory.  Arguments as in fm.hs: decimation and
--   its taps, interpolation / decimation and the resampler taps, the HALF taps of the symmetric audio
--   filter, the gain ('P.map (VG.map (* 0.2))'), and the source block size ('samples'; the seams of the
--   reference's Pipes fall at multiples of it).
gpuFmChain :: Int -> [Float] -> Int -> Int -> [Float] -> [Float] -> Float -> Int -> IO GpuFmChain
gpuFmChain decimation rfTaps interpolation decimation2 resampTaps audioHalf gain block =
    withArrayLen (map realToFrac rfTaps) $ \n1 p1 ->
    withArrayLen (map realToFrac resampTaps) $ \n2 p2 ->
    withArrayLen (map realToFrac audioHalf) $ \n3 p3 ->
    alloca $ \pp -> do
        _ <- c_chain_create pp orderAVX (fromIntegral decimation) p1 (fromIntegral n1)
                 (fromIntegral interpolation) (fromIntegral decimation2) p2 (fromIntegral n2)
                 p3 (fromIntegral n3) (realToFrac gain) (fromIntegral block) >>= check
        GpuFmChain <$> peek pp

-- | u8 IQ blocks from 'sdrStream' in, audio blocks of exactly @blockSizeOut@ floats out: same blocks,
--   bit for bit, as the five stages it replaces.  @maxBlock@ = the largest source block (a multiple of
--   the chain's block size) that will be pushed.
fmReceiverGpu :: GpuFmChain -> Int -> Int -> Pipe (VS.Vector CUChar) (VS.Vector Float) IO ()
fmReceiverGpu (GpuFmChain c) maxBlock blockSizeOut = do
    st <- lift $ alloca $ \pp -> c_stream_create pp c (fromIntegral maxBlock) (fromIntegral blockSizeOut) >>= check >> peek pp
    forever $ do
        inp   <- await
        ready <- lift $ VS.unsafeWith inp $ \ptr -> c_stream_push st ptr (fromIntegral (VS.length inp `div` 2)) >>= check
        replicateM_ (fromIntegral ready) $ do
            out <- lift $ do
                fp <- mallocForeignPtrArray blockSizeOut :: IO (ForeignPtr CFloat)
                _  <- withForeignPtr fp $ \o -> c_stream_pop st o (fromIntegral blockSizeOut) >>= check
                return $ VS.unsafeCast $ VS.unsafeFromForeignPtr0 fp blockSizeOut
            yield out
